"""Per-kernel sums of one rocprofv3 --pmc counter from a rocpd database -> CSV on stdout.
python tools/rocpd_pmc.py DB"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, counter_name, count(*), sum(counter_value), avg(counter_value) from pmc_events "
                   "group by name, counter_name order by sum(counter_value) desc").fetchall()
print("kernel,counter,calls,sum,avg")
for name, cname, n, s, a in rows:
    print(f'"{name[:110]}",{cname},{n},{s:.1f},{a:.2f}')
