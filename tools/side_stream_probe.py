"""Why did two ranks sharing one GPU go from 0.16 s to 13 s per training step with the TrainEngine's side streams on
?  One experiment that needs no second GPU and no torch.distributed:

  A  one process, side streams on                (the benchmarked single-GPU schedule)
  B  one process, side streams off               (M355_NO_WGRAD_STREAM=1 M355_NO_HEAD_STREAM=1)
  C  TWO independent processes on the same GPU, side streams on, started together, NO process group
  D  two independent processes, side streams off
  E  one process, TWO THREADS each with its own TrainEngine and side streams on (same HIP context / same queues' owner)
  F  two processes AS TWO RANKS of a gloo process group (the round-2 rehearsal itself): bucketed gradient all-reduce under
     backward, side streams forced on (M355_SIDE_STREAMS=1)
  G  the same two ranks, side streams off (what a multi-rank job runs today)

If C collapses like the round-2 rehearsal did and E does not, the cause is the time-slicing of hardware queues between
PROCESSES (each cross-stream event wait ends up waiting for the other process's queue quantum), not torch.distributed and
not the streams themselves -- and one rank per GPU (config 4 on an 8-GPU node) never shares a device between processes.
Usage: python tools/side_stream_probe.py            (parent: runs A-E and prints one table)
       python tools/side_stream_probe.py child <steps> [barrier_file]"""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SCALE, B, S = "n", 16, 320


def run_steps(steps, tag, out, gate=None, ddp=False):
    import numpy as np
    import torch
    if ddp:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    from defectdetection_viaobjectdetection_amd.loss import SegCriterion
    from defectdetection_viaobjectdetection_amd.spec import init_state_dict
    from defectdetection_viaobjectdetection_amd.train_engine import TrainEngine
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        eng = TrainEngine(SCALE, 1, (S, S), B)
        eng.load_state_dict(init_state_dict(SCALE, 1, seed=0))
        rng = np.random.default_rng(0)
        imgs = torch.from_numpy(rng.integers(0, 255, (B, S, S, 3), dtype=np.uint8)).to(dev)
        n = 2 * B
        boxes = torch.tensor(np.stack([rng.uniform(.3, .7, n), rng.uniform(.3, .7, n), rng.uniform(.1, .3, n), rng.uniform(.1, .3, n)], 1),
                             dtype=torch.float32)
        masks = torch.zeros(B, S // 4, S // 4, device=dev)
        masks[:, 20:40, 20:40] = 1
        batch = {"batch_idx": torch.arange(B).repeat_interleave(2).float(), "cls": torch.zeros(n), "bboxes": boxes, "masks": masks}
        crit = SegCriterion(1, (S, S))
        reducer = None
        if ddp:
            from defectdetection_viaobjectdetection_amd.sharding import GradBucketReducer
            reducer = GradBucketReducer(eng.flat_grads, eng.grad_spans(), bucket_bytes=1 << 20)

        def step():
            prep = crit.prepare(batch, B, dev)
            raw, protos = eng.forward(imgs)
            items, d_raw, d_protos = crit(raw, protos, prep, 128.0)
            if reducer is not None:
                reducer.reset()
            eng.backward(d_raw, d_protos, on_ready=reducer.mark_ready if reducer is not None else None)
            if reducer is not None:
                reducer.finish()
            return float(items.sum())
        for _ in range(2):
            step()
        stream.synchronize()
        if gate is not None:
            gate()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        stream.synchronize()
        out[tag] = (time.perf_counter() - t0) / steps * 1e3


def child(steps, barrier_file, ddp=False):
    def gate():
        if not barrier_file:
            return
        open(barrier_file + f".{os.getpid()}", "w").close()
        d, base = os.path.dirname(barrier_file), os.path.basename(barrier_file)
        t_end = time.time() + 120
        while len([f for f in os.listdir(d) if f.startswith(base + ".")]) < 2 and time.time() < t_end:
            time.sleep(0.005)
    out = {}
    run_steps(steps, "ms", out, None if ddp else gate, ddp)
    print(f"CHILD_MS {out['ms']:.3f}", flush=True)


def spawn(n_proc, side, steps, tag, ddp=False):
    env = dict(os.environ)
    for k in ("M355_NO_WGRAD_STREAM", "M355_NO_HEAD_STREAM", "M355_SIDE_STREAMS"):
        env.pop(k, None)
        if not side and k != "M355_SIDE_STREAMS":
            env[k] = "1"
    if ddp and side:
        env["M355_SIDE_STREAMS"] = "1"
    bar = f"/tmp/ssp_{tag}_{os.getpid()}" if n_proc > 1 else ""
    ps = []
    for r in range(n_proc):
        e = dict(env)
        if ddp:
            e.update(RANK=str(r), WORLD_SIZE=str(n_proc), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29611 + (ord(tag) % 7)))
        ps.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", str(steps), bar] + (["ddp"] if ddp else []), env=e,
                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    res = []
    for p in ps:
        o, _ = p.communicate(timeout=900)
        ms = [float(l.split()[1]) for l in o.splitlines() if l.startswith("CHILD_MS")]
        res.append(ms[0] if ms else float("nan"))
        if not ms:
            print(o[-1500:])
    if bar:
        for f in os.listdir("/tmp"):
            if f.startswith(os.path.basename(bar)):
                os.remove(os.path.join("/tmp", f))
    return res


def main():
    steps = 20
    rows = [("A one process, side streams on", spawn(1, True, steps, "A")),
            ("B one process, side streams off", spawn(1, False, steps, "B")),
            ("C two processes, side streams on", spawn(2, True, steps, "C")),
            ("D two processes, side streams off", spawn(2, False, steps, "D"))]
    # E: two threads in ONE process (this one), each with its own engine and side streams
    out, ready = {}, threading.Barrier(2)
    ts = [threading.Thread(target=run_steps, args=(steps, f"t{i}", out, ready.wait)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    rows.append(("E two threads in one process, side streams on", [out.get("t0", float("nan")), out.get("t1", float("nan"))]))
    rows.append(("F two gloo ranks (DP, 1 MiB buckets), side streams on", spawn(2, True, 6, "F", ddp=True)))
    rows.append(("G two gloo ranks (DP, 1 MiB buckets), side streams off", spawn(2, False, 6, "G", ddp=True)))
    print(f"YOLOv8{SCALE}-seg training step (forward + loss + backward), batch {B} @{S}: ms per step, per worker")
    for name, r in rows:
        print(f"  {name:48s} " + "  ".join(f"{v:9.2f}" for v in r))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else "", ddp=len(sys.argv) > 4 and sys.argv[4] == "ddp")
    else:
        main()
