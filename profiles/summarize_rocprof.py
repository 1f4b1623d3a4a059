#!/usr/bin/env python3
"""rocprofv3 results.db (rocpd sqlite) -> per-kernel summary CSV (name, calls, total_us, avg_us, pct).

    python profiles/summarize_rocprof.py gpurun_out/prof_x/x_results.db profiles/r01_x_kernel_stats.csv
"""
import csv
import sqlite3
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', n], capture_output=True, text=True).stdout.strip() or n
    except Exception:
        return n


def main(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'calls', 'total_us', 'avg_us', 'pct'])
        for name, calls, total, avg, pct in rows:
            w.writerow([demangle(name), calls, f'{total:.3f}', f'{avg:.3f}', f'{pct:.2f}'])
    print(f'wrote {out}: {len(rows)} kernels')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
