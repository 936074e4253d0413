#!/usr/bin/env python3
"""profiles/traffic_calibration.json from the two rocprofv3 --pmc passes over profiles/micro/traffic_calib.hip:
bytes a kernel REALLY moved / (counter x 1024), per access form.  make_pmc_json.py applies the factors.

  python profiles/traffic_calib.py gpurun_out/traffic_calib [out.json]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic_calibration.json")
NBYTES = 256 << 20
FORMS = {   # substring of the demangled kernel name -> (key, bytes read, bytes written)
    "read_kernel<unsigned int>": ("read_dword", NBYTES, 0),
    "read_kernel<unsigned int __vector(2)>": ("read_dwordx2", NBYTES, 0),
    "read_kernel<unsigned int __vector(4)>": ("read_dwordx4", NBYTES, 0),
    "write_kernel<unsigned int, false>": ("write_dword", 0, NBYTES),
    "write_kernel<unsigned int __vector(2), false>": ("write_dwordx2", 0, NBYTES),
    "write_kernel<unsigned int __vector(4), false>": ("write_dwordx4", 0, NBYTES),
    "write_kernel<unsigned int, true>": ("write_dword_nt", 0, NBYTES),
    "write_kernel<unsigned int __vector(4), true>": ("write_dwordx4_nt", 0, NBYTES),
    "rw_dword_rows_kernel": ("rw_dword_rows", NBYTES, NBYTES),
}
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            for sub, (key, _, _) in FORMS.items():
                if sub in name.replace("ext_vector_type", "__vector"):
                    vals[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {"bytes_per_kernel": NBYTES, "note": "factor = bytes really moved / (counter value x 1024); mean over the repetitions after the first",
       "forms": {}}
for sub, (key, rd, wr) in FORMS.items():
    e = {}
    for cn, true_bytes in (("FETCH_SIZE", rd), ("WRITE_SIZE", wr)):
        v = vals[key].get(cn)
        if not v:
            continue
        tail = v[1:] if len(v) > 1 else v
        mean = sum(tail) / len(tail)
        e[cn + "_KiB"] = mean
        if true_bytes and mean > 0:
            e[cn + "_factor"] = true_bytes / (mean * 1024.0)
    res["forms"][key] = e
    print(key, e)
with open(out_path, "w") as fh:
    json.dump(res, fh, indent=1, sort_keys=True)
print("wrote", out_path)
