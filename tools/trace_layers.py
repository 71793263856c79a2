#!/usr/bin/env python
"""Per-layer timing table from a rocprofv3 kernel trace of bench.py (dev tool)."""
import csv, sys
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('pack_weights')]
step = rows[idx[-2]:]
n, ci, B, T = 12, 24, 64, 16384
layers = []
for i in range(n): layers.append(((1 if i == 0 else i * ci), (i + 1) * ci, 15, T >> i))
layers.append((n * ci, n * ci, 15, T >> n))
for j in range(n): layers.append(((2 * n * ci if j == 0 else (2 * (n - j) + 1) * ci), (n - j) * ci, 5, T >> (n - 1 - j)))
fl = lambda l: 2.0 * B * l[3] * l[0] * l[1] * l[2]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
convs = [r for r in step if 'conv_mfma' in r['Kernel_Name']]
wgs = [r for r in step if 'wgrad_mfma' in r['Kernel_Name']]
tf = 0; td = 0; tw = 0
def show(tag, i, r):
    d = dur(r)
    print(f" L{i:2d} {tag} {r['Kernel_Name'][5:42]:38s} grid {int(r['Grid_Size_X'])//256:>6d},{r['Grid_Size_Y']},{r['Grid_Size_Z']} vgpr {r['VGPR_Count']:>3s}+{r['Accum_VGPR_Count']:>3s} {d:8.1f} us {fl(layers[i])/d/1e6:7.1f} TF")
    return d
for i in range(25):
    tf += show('fwd  ', i, convs[i])
for k, i in enumerate(range(24, 0, -1)):
    td += show('dgrad', i, convs[25 + k])
for k, i in enumerate(range(24, -1, -1)):
    tw += show('wgrad', i, wgs[k])
print(f"fwd {tf:.0f} us  dgrad {td:.0f} us  wgrad {tw:.0f} us  total mfma {tf+td+tw:.0f} us; step span {(int(step[-1]['End_Timestamp'])-int(step[0]['Start_Timestamp']))/1e3:.0f} us")
other = {}
for r in step:
    nm = r['Kernel_Name'].split('(')[0][:44]
    if 'conv_mfma' in nm or ('wgrad' in nm and 'reduce' not in nm): continue
    other[nm] = other.get(nm, 0) + dur(r)
for k_, v in sorted(other.items(), key=lambda x: -x[1]): print(f"  {k_:46s} {v:9.1f} us")
