import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import vectordb_amd
from oracle import pyoracle
import test_gpu_c4_shape as t
vdb, ora = vectordb_amd.api(), pyoracle.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
x = t.unit_rows(n, 11); cat, tier = t.columns(n)
def lists(g):
    out = []
    for ow, of, nb in g["layers"]:
        out.append({int(o): nb[int(of[i]):int(of[i+1])].tolist() for i, o in enumerate(ow)})
    return out
# index only
ho = ora.HNSWIndex(1, 16, 64, 16, t.D, seed=0xC4); hp = vdb.HNSWIndex(1, 16, 64, 16, t.D, seed=0xC4, arith=vdb.Arith.Strict)
for i in range(n):
    ho.insert(i + 1, x[i]); hp.insert(i + 1, x[i])
lo, lp = lists(ho.export_graph()), lists(hp.export_graph())
print("index only: same", lo == lp)
if lo != lp:
    for l, (a, b) in enumerate(zip(lo, lp)):
        bad = [k for k in a if a[k] != b.get(k)]
        print(" layer", l, "differing lists", len(bad), bad[:5])
        for k in bad[:2]: print("   ", k, a[k], b.get(k))
# segment, find first insert at which graphs differ
so = ora.Segment(ora.HNSWIndex(1, 16, 64, 16, t.D, seed=0xC4, filter_mode=1)); sp = vdb.Segment(vdb.HNSWIndex(1, 16, 64, 16, t.D, seed=0xC4, filter_mode=1, arith=vdb.Arith.Strict))
def pay(a, i):
    p = a.Payload(); p.set("cat", a.PayloadValue.Int(int(cat[i]))); p.set("tier", a.PayloadValue.Int(int(tier[i]))); return p
for i in range(n):
    so.insert(x[i], pay(ora, i)); sp.insert(x[i], pay(vdb, i))
    if i % 50 == 49 or i < 20:
        a, b = lists(so.hnsw().export_graph()), lists(sp.hnsw().export_graph())
        if a != b:
            print("segment differs after insert", i)
            for l, (aa, bb) in enumerate(zip(a, b)):
                bad = [k for k in aa if aa[k] != bb.get(k)]
                print(" layer", l, "differing lists", len(bad), bad[:8])
                for k in bad[:3]: print("   ", k, aa[k], bb.get(k))
            break
else:
    print("segment same")
