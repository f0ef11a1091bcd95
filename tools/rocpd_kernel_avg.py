import sqlite3,sys
db=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd=[t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
for r in db.execute(f"select s.kernel_name, count(*), avg(d.end-d.start)/1e6 from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 3 desc"): print("  %-60s %5d %8.3f ms"%(r[0][:60],r[1],r[2]))
