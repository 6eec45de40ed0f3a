import csv, sys, collections
def load(path):
    rows=list(csv.DictReader(open(path)))
    return rows
sq=load('gpurun_out/pmc/sq_counter_collection.csv')
print(sq[0].keys())
# group by dispatch id
disp=collections.OrderedDict()
for r in sq:
    d=disp.setdefault(int(r['Dispatch_Id']),{'name':r['Kernel_Name'],'grid':r.get('Grid_Size')})
    d[r['Counter_Name']]=d.get(r['Counter_Name'],0)+float(r['Counter_Value'])
fe={}
for r in load('gpurun_out/pmc/fetch_counter_collection.csv'):
    fe[int(r['Dispatch_Id'])]=fe.get(int(r['Dispatch_Id']),0)+float(r['Counter_Value'])
wr={}
for r in load('gpurun_out/pmc/write_counter_collection.csv'):
    wr[int(r['Dispatch_Id'])]=wr.get(int(r['Dispatch_Id']),0)+float(r['Counter_Value'])
ids=list(disp)
# last step: find last stem
stems=[i for i in ids if 'stem_kernel' in disp[i]['name']]
start=stems[-1]
print("disp kernel grid wave_cyc busy wait_any% wait_inst% active% mfma_busy/busy valu/wave LDSconf fetchMB(x2) writeMB")
for i in ids:
    if i<start: continue
    d=disp[i]; n=d['name']
    n=n[n.find('conv_igemm_kernel')+17:][:16] if 'conv_igemm' in n else n[n.find('N_1')+4:][:16] if '_ZN' in n else n[:30]
    wc=d.get('SQ_WAVE_CYCLES',0); 
    if wc==0: continue
    print(f"{i:5d} {n:18s} {d['grid']:>8s} wc={wc:.3g} busy={d.get('SQ_BUSY_CYCLES',0):.3g} wait={100*d.get('SQ_WAIT_ANY',0)/wc:5.1f} winst={100*d.get('SQ_WAIT_INST_ANY',0)/wc:5.1f} act={100*d.get('SQ_ACTIVE_INST_ANY',0)/wc:5.1f} mfma={d.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.3g} valu={d.get('SQ_INSTS_VALU',0):.3g} ldsc={d.get('SQ_LDS_BANK_CONFLICT',0):.3g} F={2*fe.get(i,0)/1024:.1f}MB W={wr.get(i,0)/1024:.1f}MB")
