"""Merge gpurun_out/<tag>/traffic_entry.json files (written on the GPU box by scripts/summarize_profile.py) into
profiles/traffic.json, keyed by workload:outputs:B<draws>.  bench.py reports the entry whose kernel matches the one it
launched as roofline.traffic, with traffic_source = the entry's source + commit."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles", "traffic.json")
db = json.load(open(dst)) if os.path.exists(dst) else {}
db = {k: v for k, v in db.items() if isinstance(v, dict)}  # drop the round-1 hand-written scalar entries
for f in sys.argv[1:]:
    e = json.load(open(f))
    key = e.pop("key")
    e["source"] = f"{e['source']} @ {e.pop('commit')}"
    db[key] = e
    print(key, "->", e)
json.dump(db, open(dst, "w"), indent=1, sort_keys=True)
