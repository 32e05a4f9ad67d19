"""Per-kernel timeline of ONE static-shape mini-batch step (BASELINE config 4), from a rocprofv3
kernel trace of the eager form of the step:
  PYGAMD_CAPTURE=0 rocprofv3 --kernel-trace -d OUT -o t --output-format csv -- \
      python bench.py --mode minibatch --capture --steps 6 --warmup 2 [--scale S]
  python scripts/minibatch_step_trace.py OUT/t_kernel_trace.csv
Prints the kernels of the last-but-one training step (one `slots_seed` launch to the next), in order, with their durations and
the idle gap in front of each, plus totals by name."""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
seeds = [i for i, r in enumerate(rows) if 'slots_seed' in r[2]]
# the last seed-to-seed interval that holds a training step (the statistics recount behind the
# timed region samples again, without a model)
pairs = [(a, b) for a, b in zip(seeds, seeds[1:])
         if any('sage_fused' in r[2] for r in rows[a:b])]
a, b = pairs[-2] if len(pairs) > 1 else pairs[-1]
step = rows[a:b]
prev_end = step[0][0]
tot = collections.OrderedDict()
busy = 0
for s, e, k in step:
    name = k.split('(')[0].replace('void ', '')[:70]
    print(f'{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  {name}')
    prev_end = e
    tot.setdefault(name, [0, 0.0])
    tot[name][0] += 1
    tot[name][1] += (e - s) / 1e3
    busy += e - s
print(f'\n{len(step)} launches, busy {busy / 1e3:.1f} us, span {(step[-1][1] - step[0][0]) / 1e3:.1f} us')
for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f'{t:8.1f} us  x{n:2d}  {k}')
