"""Category + phase breakdown of the last full step in a rocprofv3 kernel trace."""
import collections, csv, sys

def cat(n):
    if 'dfine::' in n[:16]: return 'dfine HIP (this repo)'
    if 'igemm' in n or 'ck::' in n or '_ZN2ck' in n or 'naive_conv' in n: return 'MIOpen conv'
    if n.startswith('Cijk'): return 'hipBLASLt GEMM'
    if 'SubTensorOp' in n or 'MIOpen' in n: return 'MIOpen misc (tensor ops, BN)'
    if 'attn' in n or 'bwd_kernel' in n: return 'attention (aotriton)'
    if 'elementwise' in n or 'FillFunctor' in n or 'fillBuffer' in n: return 'ATen elementwise/fill'
    if 'reduce_kernel' in n: return 'ATen reduce'
    return 'other ATen'

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'msda_fwd' in r['Kernel_Name']][::4]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
sel = rows[marks[which]:marks[which + 1]] if which + 1 != 0 else rows[marks[which]:]
names = [r['Kernel_Name'] for r in sel]
dur = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in sel]
wall = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e6
print(f"step: wall {wall:.2f} ms, busy {sum(dur)/1e6:.2f} ms, {len(sel)} launches")
agg = collections.defaultdict(lambda: [0, 0])
for n, d in zip(names, dur):
    agg[cat(n)][0] += d; agg[cat(n)][1] += 1
for c, (d, k) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"  {c:30s} {d/1e6:8.2f} ms {100*d/sum(dur):5.1f}%  {k:5d} launches")
def first(sub):
    for i, n in enumerate(names):
        if sub in n: return i
i_cost, i_bwd, i_opt = first('match_cost'), first('msda_bwd'), first('grad_sumsq')
for a, b, label in ((0, i_cost, 'decoder fwd (from first msda_fwd)'), (i_cost, i_bwd, 'criterion fwd+bwd'), (i_bwd, i_opt, 'backward'), (i_opt, len(names), 'optimizer + next backbone/encoder fwd')):
    print(f"  {label:40s} launches {b-a:5d}  busy {sum(dur[a:b])/1e6:7.2f} ms")
