"""avg of one counter per kernel from a rocprofv3 --pmc pass (rocpd SQLite): python scripts/pmc_quick.py <dir or db> <COUNTER> [more counters]"""
import os
import sqlite3
import sys


def find_db(path):
    if os.path.isfile(path):
        return path
    for root, _, files in os.walk(path):
        for f in files:
            if f.endswith("_results.db"):
                return os.path.join(root, f)
    raise SystemExit(f"no *_results.db under {path}")


c = sqlite3.connect(find_db(sys.argv[1]))
for ctr in sys.argv[2:]:
    for k, n, avg in c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name order by 3 desc", (ctr,)):
        name = k.replace("vima::(anonymous namespace)::", "vima::").replace("void ", "").split("(")[0][:70]
        print(f"  {ctr} {name}: {n} dispatches, avg {avg:.1f}")
