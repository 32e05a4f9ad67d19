"""Turn a rocprofv3 --kernel-trace --stats CSV into a short markdown table under profiles/."""
import csv
import re
import sys


def short(n):
    n = n.replace('void ', '')
    if n.startswith('Cijk'):
        m = re.search(r'_(MT\d+x\d+x\d+)_', n)
        return 'rocBLAS/hipBLASLt ' + n[5:14] + ' ' + (m.group(1) if m else '')
    if 'rocprim' in n:
        m = re.search(r'detail::(radix_sort\w+|partition_impl|transform_impl)', n)
        return 'rocprim ' + (m.group(1) if m else n[:60])
    return n[:120]


def main(src, dst, title, steps):
    rows = list(csv.DictReader(open(src)))
    with open(dst, 'w') as f:
        f.write(f'# {title}\n\n')
        f.write('| kernel | calls | total ms | avg ms | ms/step | % |\n|---|---:|---:|---:|---:|---:|\n')
        for r in rows[:28]:
            tot = int(r['TotalDurationNs']) / 1e6
            f.write(f"| `{short(r['Name'])}` | {r['Calls']} | {tot:.2f} | "
                    f"{float(r['AverageNs']) / 1e6:.4f} | {tot / steps:.3f} | {r['Percentage']} |\n")
        total = sum(int(r['TotalDurationNs']) for r in rows) / 1e6
        f.write(f'\nSum of all kernels: {total:.1f} ms over {steps} steps '
                f'({total / steps:.2f} ms/step).\n')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]))
