"""Top kernels from a rocprofv3 rocpd database: python tools/rocpd_top.py <dir-or-db> [n]"""
import glob, os, sqlite3, sys
src = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
db = src if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))[0]
c = sqlite3.connect(db)
tot = c.execute("select sum(end-start) from kernels").fetchone()[0]
print(f"total kernel time {tot/1e6:.3f} ms")
for r in c.execute("select name, count(*), avg(end-start), sum(end-start) from kernels group by name order by 4 desc limit ?", (n,)):
    print(r[0].replace("void ", "")[:92].ljust(92), str(r[1]).rjust(6), f"{r[2]/1e3:9.2f}us {100*r[3]/tot:6.2f}%")
