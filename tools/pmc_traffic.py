"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.
gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly half the bytes of wide
coalesced streaming reads -> doubled; WRITE_SIZE is exact for 16-byte-per-lane stores; both are in KiB."""
import csv, json, re, sys, collections

LABELS = {
    "c2f_c32_kernel": "c2f_c32<8x16px>", "proto_phase_wreg_kernel": "proto_phase_wreg<8x16px>",
    "conv3x3_s2c64_cv1_kernel": "conv3x3_s2c64<8x8px>+1x1", "head_tail_kernel": "head_tail<128px>",
    "PCfg<4, 9, 7, true": "bneck_pair<128ch>", "PCfg<2, 8, 6, true": "bneck_pair<64ch>", "PCfg<2, 0, 4, false": "conv3x3_planes<64ch,rows>",
    "PCfgILi4ELi9ELi7ELb1": "bneck_pair<128ch>", "PCfgILi2ELi8ELi6ELb1": "bneck_pair<64ch>", "PCfgILi2ELi0ELi4ELb0": "conv3x3_planes<64ch,rows>",
    "conv3x3_halo_kernel<4, 4, 2, 2>": "conv3x3_halo<128ch>", "conv3x3_halo_kernel<4, 2, 1, 4>": "conv3x3_halo<64ch>",
    "conv3x3_halo_kernel<4, 4, 2, 4>": "conv3x3_halo<128ch>", "conv3x3_halo_kernel<4, 2, 1, 8>": "conv3x3_halo<64ch>",
    "conv_igemm_kernel<4, 4, 2, 2, 3>": "conv_igemm<128x128,k3>", "conv_igemm_kernel<4, 4, 2, 2, 1>": "conv_igemm<128x128,k1>",
    "conv_igemm_kernel<4, 4, 2, 2, 2>": "conv_igemm<128x128,k2,phase+1x1>", "conv_igemm_kernel<4, 2, 1, 4, 2>": "conv_igemm<64x128,k2,phase>",
    "conv_igemm_kernel<4, 2, 1, 4, 3>": "conv_igemm<64x128,k3>", "conv_igemm_kernel<4, 2, 1, 4, 1>": "conv_igemm<64x128,k1>",
    "conv_igemm_kernel<2, 4, 1, 4, 3>": "conv_igemm<32x256,k3>", "conv_igemm_kernel<2, 4, 1, 4, 1>": "conv_igemm<32x256,k1>",
    "conv3x3_m32_kernel<128, 2>": "conv3x3_m32<128ch,8x16px>", "conv3x3_m32_kernel<64, 1>": "conv3x3_m32<64ch,8x16px>",
    "conv3x3_m32_kernel<64, 2>": "conv3x3_m32<64ch,16x16px>",
    "conv3x3_wide_kernel": "conv3x3_wide<128ch,16x16px>", "conv3x3_slab_kernel": "conv3x3_slab<64ch,rows>", "conv3x3_c32_kernel": "conv3x3_c32<32ch,16x16px>",
    "stem_s2c32_cv1_kernel": "stem+conv3x3_s2c32<8x16px>+1x1", "stem_s2c32_cv1_v2_kernel": "stem+conv3x3_s2c32<8x16px>+1x1", "conv_s2c32_cv1_kernel": "conv3x3_s2c32<8x16px>+1x1",
    "stem_rows_kernel": "stem_conv<k3s2,u8,mfma>", "augment_kernel": "augment",
    "stem_kernel": "stem_conv<k3s2,u8,mfma>", "head_decode_kernel": "head_decode", "sppf_pool_kernel": "sppf_pool",
    "upsample2x_kernel": "upsample2x", "nms_kernel": "nms", "proto_masks_kernel": "proto_masks",
}
MANGLED = {"ILi4ELi4ELi2ELi2EEEvNS_8ConvArgsEiiii": "conv3x3_halo<128ch>", "ILi4ELi2ELi1ELi4EEEvNS_8ConvArgsEiiii": "conv3x3_halo<64ch>"}


def label(name):
    # conv_igemm_kernel<MT, NT, WCH, WPX, KS, EPI>: the epilogue selector (0 plain, 1 decode, 2 phase + 1x1) is not part of the label
    name = re.sub(r"(conv_igemm_kernel<\d+, \d+, \d+, \d+, \d+), \d+>", r"\1>", name)
    mw = re.search(r"conv1x1_wreg_kernel<(\d+), (\d+)", name)      # <K, CB, PB, SPLIT> since round 4 (<K, CB> before)
    if mw:
        return f"conv1x1_wreg<K{mw.group(1)},{int(mw.group(2)) * 32}ch>"
    for k, v in LABELS.items():
        if k in name:
            return v
    for k, v in MANGLED.items():
        if k in name:
            return v
    for k in ("conv3x3_wide", "conv3x3_slab", "conv3x3_c32", "stem_rows", "sppf_pool", "upsample2x", "proto_masks", "nms_kernel", "head_decode", "stem_kernel"):
        if k in name:
            return LABELS.get(k + "_kernel", LABELS.get(k, k))
    return None


def load(path, counter):
    per = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        lb = label(r["Kernel_Name"])
        if lb is None:
            continue
        per[lb][0] += float(r["Counter_Value"])
        per[lb][1].add(r["Dispatch_Id"])
    return {k: (v[0], len(v[1])) for k, v in per.items()}


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        n = max(nf, nw, 1)
        out[k] = {"launches": n, "fetch_bytes_per_launch": 2.0 * f * 1024 / max(nf, 1), "write_bytes_per_launch": w * 1024 / max(nw, 1)}
        out[k]["hbm_bytes_per_launch"] = out[k]["fetch_bytes_per_launch"] + out[k]["write_bytes_per_launch"]
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH doubled per the gfx950 correction; "
                       "bench.py --serial --steps 2 --warmup 1 --batch 32", "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        print(f"{k:28s} launches {v['launches']:4d}  HBM/launch {v['hbm_bytes_per_launch'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
