#!/usr/bin/env python3
"""HBM traffic per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE passes, as the
TCC slot budget demands).  Units/corrections per MI355X_MICROARCH.md section HBM: the counters are in KB; on gfx950 FETCH_SIZE
reports half the bytes of wide coalesced reads, so both the raw and the doubled read figure are recorded; WRITE_SIZE is taken
as is (it matches the known brick store byte-for-byte here).  usage: traffic_json.py fetch.db write.db out.json"""
import json, sqlite3, sys
def mean(db, counter):
    con = sqlite3.connect(db); cur = con.cursor()
    cur.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name=? group by kernel_name", (counter,))
    rows = cur.fetchall(); con.close()
    return {n.replace("(anonymous namespace)::", "").replace("void ", "").split("<")[0].split("(")[0]: (v, k) for n, v, k in rows}
f, w = mean(sys.argv[1], "FETCH_SIZE"), mean(sys.argv[2], "WRITE_SIZE")
import hashlib, os
def kernel_sources_sha():
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "volumetric-particles-for-unity_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn in ("fill.hip", "fill_generic.hip", "fill_kernels.h", "raymarch.hip", "raymarch_generic.hip", "raymarch_kernels.h", "bin.hip", "vpfx_internal.h"):      # the kernels (same fingerprint as bench.py)
            h.update(fn.encode() + b"\0" + open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]
out = {"kernel_sources_sha": kernel_sources_sha()}
for k in ("k_fill", "k_fill_lds", "k_raymarch"):
    if k in f and k in w:
        fk, wk = f[k][0] * 1024.0, w[k][0] * 1024.0
        out[k] = {"FETCH_SIZE_bytes_raw": fk, "WRITE_SIZE_bytes": wk, "launches_averaged": f[k][1],
                  "traffic_bytes": 2.0 * fk + wk, "note": "traffic = 2*FETCH_SIZE (gfx950 half-count correction) + WRITE_SIZE"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
