"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output) into a per-kernel CSV:
name, grid, calls, total_us, avg_us, min_us, max_us, pct.   python tools/rocpd_stats.py DB [> out.csv]"""
import sqlite3
import sys


def main(path, min_calls=1):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute(
        "select name, grid_x/workgroup_x, grid_y/workgroup_y, grid_z/workgroup_z, workgroup_x, count(*), "
        "sum(end-start), avg(end-start), min(end-start), max(end-start), vgpr_count, accum_vgpr_count, lds_size "
        "from kernels group by name, grid_x, grid_y, grid_z order by sum(end-start) desc").fetchall()
    total = sum(r[6] for r in rows) or 1
    print("kernel,blocks,threads_per_block,calls,total_us,avg_us,min_us,max_us,pct,vgpr,agpr,lds_bytes")
    for r in rows:
        if r[5] < min_calls:
            continue
        name = r[0].replace('"', "'")
        print(f'"{name}",{r[1]}x{r[2]}x{r[3]},{r[4]},{r[5]},{r[6] / 1e3:.1f},{r[7] / 1e3:.2f},{r[8] / 1e3:.2f},'
              f'{r[9] / 1e3:.2f},{100.0 * r[6] / total:.2f},{r[10]},{r[11]},{r[12]}')


if __name__ == "__main__":
    main(sys.argv[1])
