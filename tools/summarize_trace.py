"""Reduce a rocprofv3 --kernel-trace CSV to a per-step kernel summary.

A benchmark step ends with ia::k_finalize (not k_finalize_part); everything between two consecutive
k_finalize completions is one step.  MIOpen find-mode trials (first steps) are
excluded by summarising only the last `--steps` steps.

    python tools/summarize_trace.py <kernel_trace.csv> --steps 5 > profiles/xxx.txt
"""
import argparse
import csv
import collections
import re
import sys

ap = argparse.ArgumentParser()
ap.add_argument('trace')
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--marker', default='ia::k_finalize(')
ap.add_argument('--top', type=int, default=45)
args = ap.parse_args()

rows = []
with open(args.trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
marks = [i for i, r in enumerate(rows) if args.marker in r[2]]
if len(marks) < args.steps + 1:
    sys.exit('only %d markers' % len(marks))
lo, hi = marks[-args.steps - 1] + 1, marks[-1] + 1
sel = rows[lo:hi]
wall = (sel[-1][1] - rows[marks[-args.steps - 1]][1]) / 1e6 / args.steps


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'at::native::', '', n)
    return n[:110]


agg = collections.OrderedDict()
for s, e, n in sel:
    k = short(n)
    c = agg.setdefault(k, [0, 0])
    c[0] += 1
    c[1] += e - s
busy = sum(v[1] for v in agg.values()) / 1e6 / args.steps
print('steps summarised: %d   wall per step: %.3f ms   kernel-busy per step: %.3f ms   launches '
      'per step: %.1f' % (args.steps, wall, busy, len(sel) / args.steps))
print('%-112s %8s %10s %10s %6s' % ('kernel', 'calls/st', 'avg us', 'ms/step', '%'))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
    print('%-112s %8.1f %10.1f %10.3f %6.2f' % (k, c / args.steps, t / c / 1e3, t / 1e6 / args.steps,
                                                 100.0 * t / 1e6 / args.steps / busy))

# roll-up by kind: who owns the time
KINDS = [('library GEMM (hipBLASLt / rocBLAS)', r'^Cijk_'),
         ('MIOpen / CK convolution', r'igemm_|miopen|Sp3Asm|ck::|kernel_grouped_conv|batched_transpose|SubTensorOp|MIOpen'),
         ('Winograd transforms (ia::k_wino_*)', r'ia::k_wino_'),
         ('other kernels of this library (ia::)', r'ia::'),
         ('torch copies / fills', r'direct_copy|fillBuffer|copyBuffer|FillFunctor'),
         ('torch reductions', r'reduce_kernel'),
         ('torch optimizer / clip (multi_tensor)', r'multi_tensor_apply'),
         ('torch elementwise + rest', r'.')]
kinds = collections.OrderedDict((k, [0, 0]) for k, _ in KINDS)
for k, (c, t) in agg.items():
    for name, pat in KINDS:
        if re.search(pat, k):
            kinds[name][0] += c
            kinds[name][1] += t
            break
print()
print('%-112s %8s %10s %10s %6s' % ('by kind', 'calls/st', '', 'ms/step', '%'))
for name, (c, t) in kinds.items():
    print('%-112s %8.1f %10s %10.3f %6.2f' % (name, c / args.steps, '', t / 1e6 / args.steps,
                                               100.0 * t / 1e6 / args.steps / busy))
