import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import plan
d=sys.argv[1]; pre=sys.argv[2]
rows=list(csv.DictReader(open(f'{d}/{pre}_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if n.startswith('pack_weights_kernel')]
seq=rows[idx[-2]:]
layers = plan.conv_layers(12, 24); B=64
Ls=[]
for i,(n,ci,co,t) in enumerate(layers):
    if n.startswith('encoder'): L=16384>>int(n.split('.')[1])
    elif n.startswith('middle'): L=16384>>12
    else: L=16384>>(11-int(n.split('.')[1]))
    Ls.append(L)
fw=[r for r in seq if 'conv_h3' in r['Kernel_Name'] or 'conv_mfma' in r['Kernel_Name'] or 'conv_first' in r['Kernel_Name']]
wg=[r for r in seq if ('wgrad_h3_kernel' in r['Kernel_Name'] or 'wgrad_h3d_kernel' in r['Kernel_Name'] or 'wgrad_mfma' in r['Kernel_Name'])]
dur=lambda r:(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
fwd=fw[:25]; dg=fw[25:][::-1]   # dgrad in reverse order: layers 24..1
print("layer              L    cin cout | fwd us  (TF)  hbm-floor | dgrad us (TF) | wgrad us (TF)")
tf=lambda fl,us: fl/us/1e6
for i,(n,ci,co,t) in enumerate(layers):
    L=Ls[i]; fl=2.0*B*L*ci*co*t
    f=dur(fwd[i]); 
    d_=dur(dg[i-1]) if i>0 else 0
    w=dur(wg[::-1][i - (25 - len(wg))]) if i >= 25 - len(wg) else 0.0      # (the first layer's weight gradient may come from pass A's sums: no kernel)
    hb=(B*L*(ci+co)*4)/5e6  # us at 5 TB/s
    print("%-16s %5d %4d %4d | %6.1f %5.0f %6.1f | %6.1f %5.0f | %6.1f %5.0f  %s"%(n,L,ci,co,f,tf(fl,f),hb,d_,tf(fl,d_) if i>0 else 0,w,tf(fl,w) if w else 0, fwd[i]['Kernel_Name'][5:28]))
