"""Per-layer attainable time for YOLOv8s-seg b32 from the graph spec: max(FLOPs/MFMA_rate, HBM bytes/BW, LDS-DMA intake/rate)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from defectdetection_viaobjectdetection_amd.spec import conv_specs
B=32
MF=1.0e15; HBM=5.0e12; INTAKE=12.8e12
# spatial size per layer name prefix (s-seg @640)
def hw(name):
    table={'model.0':320,'model.1':160,'model.2':160,'model.3':80,'model.4':80,'model.5':40,'model.6':40,'model.7':20,'model.8':20,'model.9':20,
           'model.12':40,'model.15':80,'model.16':40,'model.18':40,'model.19':20,'model.21':20}
    for k,v in table.items():
        if name==k or name.startswith(k+'.'): return v
    if name.startswith('model.22.proto.cv1'): return 80
    if name.startswith('model.22.proto.upsample'): return 160
    if name.startswith('model.22.proto'): return 160
    l=int(name.split('.')[3]); return (80,40,20)[l]
tot=dict(flop=0,hbm=0,intake=0,best=0)
rows=[]
for s in conv_specs('s',1):
    o=hw(s.name); 
    px=B*o*o
    k=s.k*s.k if not s.transposed else 1
    cout=s.cout*(4 if s.transposed else 1); opx = px//4 if s.transposed else px
    flops=2*opx*cout*s.cin*k
    inpx = px*(s.stride**2) if not s.transposed else px//4
    hbm=(inpx*s.cin + px*s.cout)*2
    # im2col intake with 128x128 tiles: per 128px x 128ch tile, K*(128+128)*2 bytes
    K=s.cin*k
    tiles=(opx/128)*max(cout/128,1) if cout>=64 else (opx/256)
    bch=128 if cout>64 else (64 if cout>32 else 32); bpx=128 if cout>32 else 256
    tiles=(opx/bpx)*-(-cout//bch)
    intake=tiles*K*(bch+bpx)*2
    t=max(flops/MF,hbm/HBM,intake/INTAKE)
    rows.append((s.name,s.cin,s.cout,s.k,o,flops/MF*1e6,hbm/HBM*1e6,intake/INTAKE*1e6,t*1e6))
    tot['flop']+=flops/MF; tot['hbm']+=hbm/HBM; tot['intake']+=intake/INTAKE; tot['best']+=max(flops/MF,hbm/HBM)
for r in rows: print("%-26s %4d->%4d k%d @%3d  mfma %6.1f  hbm %6.1f  intake %6.1f  max %6.1f us"%r)
print({k:round(v*1e3,3) for k,v in tot.items()}, 'ms;  sum max(all3)=', round(sum(r[-1] for r in rows)/1e3,3))
