import sqlite3, collections, re, sys
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
sym={r[0]:r[1] for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
rows=list(cur.execute("select kernel_id,queue_id,stream_id,start,end,tid from rocpd_kernel_dispatch order by start"))
t0=rows[0][3]
def short(n):
    m=re.search(r'(k_[a-z_]+)(ILi(\d+))?',n); return (m.group(1)+('<%s>'%m.group(3) if m.group(3) else '')) if m else n[:20]
tend=max(r[4] for r in rows)
sel=[r for r in rows if r[3]>tend-float(sys.argv[2])*1e6 and (r[4]-r[3])>1.5e6]
print("queues used:",len(set(r[1] for r in rows)),"streams:",len(set(r[2] for r in rows)))
for k,q,s,st,en,tid in sel[:int(sys.argv[3])]:
    print("q%-3d s%-3d tid%-7d %8.1f -> %8.1f (%6.1f ms) %s"%(q,s,tid%100000,(st-t0)/1e6,(en-t0)/1e6,(en-st)/1e6,short(sym[k])))
