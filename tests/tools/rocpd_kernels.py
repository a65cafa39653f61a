import sqlite3,statistics,sys,glob
from collections import defaultdict
db=sqlite3.connect(glob.glob(sys.argv[1]+'/*.db')[0])
rs=db.execute("select name,grid_x,start,end from kernels order by start").fetchall()
g=defaultdict(list)
for n,gx,s,e in rs:
    if 'd3' in n or 'igemm' in n: g[(n[:44],gx)].append(e-s)
for k,v in g.items(): print(k, len(v), 'median', statistics.median(v), 'min', min(v))
