"""Development aid: which kernels ran on which hardware queue (rocprofv3 kernel trace csv) -- python tools/queue_map.py trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", ", ".join(rows[0].keys()))
q = collections.defaultdict(lambda: collections.Counter())
for r in rows:
    key = (r.get("Queue_Id"), r.get("Stream_Id"))
    q[key][r["Kernel_Name"].split("(")[0][:40]] += 1
for key in sorted(q):
    tot = sum(q[key].values())
    print("queue %s stream %s: %d launches: %s" % (key[0], key[1], tot, ", ".join("%s x%d" % kv for kv in q[key].most_common(8))))
