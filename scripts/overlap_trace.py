"""Do the two streams of the pipelined slot step (slots.SlotTrainer(pipeline=True)) overlap?  From a
rocprofv3 kernel trace of `bench.py --mode minibatch --capture`: for the last steps, every kernel
with its queue, start offset, duration and the kernels of the OTHER queue running at its start.
Usage: python scripts/overlap_trace.py OUT/t_kernel_trace.csv"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                     r['Kernel_Name'].split('(')[0].replace('void ', '')[:44], r.get('Queue_Id', '?')))
rows.sort()
adam = [i for i, r in enumerate(rows) if 'adam_step' in r[2]]
a, b = adam[-4], adam[-2]
win = rows[a + 1:b + 1]
t0 = win[0][0]
busy = sum(e - s for s, e, _, _ in win)
# union of the intervals
union, cur_s, cur_e = 0, None, None
for s, e, _, _ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print(f'two steps: span {(win[-1][1] - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us, '
      f'union {union / 1e3:.1f} us (overlapped {(busy - union) / 1e3:.1f} us)')
for i, (s, e, k, q) in enumerate(win):
    others = [k2 for s2, e2, k2, q2 in win if q2 != q and s2 <= s < e2]
    print(f'{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  q{q:>3s}  {k:44s} {"| " + ", ".join(others) if others else ""}')
