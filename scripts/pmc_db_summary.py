#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 sqlite output (pmc_results.db)."""
import sqlite3
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    con = sqlite3.connect(d)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pm = [t for t in tabs if 'pmc_event' in t][0]
    info = [t for t in tabs if 'info_pmc' in t][0]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]
    ks = [t for t in tabs if 'info_kernel_symbol' in t][0]
    rows = cur.execute(f"""select s.kernel_name, i.name, e.value, k.id, k.grid_size_x from {pm} e join {info} i on e.pmc_id = i.id
                           join {kd} k on e.event_id = k.event_id join {ks} s on k.kernel_id = s.id""").fetchall()
    acc = defaultdict(lambda: defaultdict(list))
    for kn, cn, v, kid, gx in rows:
        acc[kn[:90]][cn].append(v)
    print("==", d)
    for kn, cs in acc.items():
        if 'gemm' not in kn and 'Cijk' not in kn:
            continue
        print(kn, " dispatches:", len(next(iter(cs.values()))))
        for cn, vs in sorted(cs.items()):
            print(f"    {cn:40s} {sum(vs)/len(vs):16.1f}")
