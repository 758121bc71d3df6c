"""Extract the parameter-blob LAYOUT of the reference's example/c906_mobilenetv1_f16.c (run in the build container only).

The example addresses one malloc'ed blob through `params_base + <offset>` for every quantisation record and every
constant tensor (c906_mobilenetv1_f16.c:34-1886) and never initialises it.  To run it deterministically
(tests/test_ref_example.py) the bytes must be meaningful: binary16 records need scale == 1, weights need sane
values.  This script reads the offsets, shapes and layouts out of the source where it lies under /root/reference and
writes them -- numbers only, no source text -- to tests/golden/example_c906_mobilenetv1_f16_layout.json.

    python tests/golden/make_example_layout.py
"""
import json
import os
import re

SRC = "/root/reference/example/c906_mobilenetv1_f16.c"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "example_c906_mobilenetv1_f16_layout.json")


def main():
    text = open(SRC).read()
    tensors = {}
    for m in re.finditer(r"(\w+)->(\w+)(?:\[(\d)\])? = ([^;]+);", text):
        var, field, idx, val = m.group(1), m.group(2), m.group(3), m.group(4).strip()
        t = tensors.setdefault(var, {"dim": {}})
        if field == "dim":
            t["dim"][int(idx)] = int(val)
        elif field == "data":
            t["data"] = int(re.search(r"params_base \+ (\d+)", val).group(1))
        elif field == "qinfo":
            t["qinfo"] = int(re.search(r"params_base \+ (\d+)", val).group(1))
        elif field in ("dim_count", "group"):
            t[field] = int(val)
        elif field == "layout":
            t["layout"] = val.replace("CSINN_LAYOUT_", "")
    qinfo = sorted(t["qinfo"] for t in tensors.values() if "qinfo" in t)
    consts = []
    for name, t in tensors.items():
        if "data" not in t:
            continue
        shape = [t["dim"][i] for i in range(t["dim_count"])]
        consts.append({"name": name, "offset": t["data"], "shape": shape, "layout": t["layout"]})
    consts.sort(key=lambda c: c["offset"])
    malloc_bytes = int(re.search(r"params = malloc\((\d+)\)", text).group(1))
    base = int(re.search(r"csinn_\(params \+ (\d+)\)", text).group(1))
    layout = {"source": "example/c906_mobilenetv1_f16.c", "malloc_bytes": malloc_bytes, "params_base": base,
              "input_bytes": 224 * 224 * 3 * 2, "qinfo_offsets": qinfo, "consts": consts}
    for c in consts:
        n = 1
        for d in c["shape"]:
            n *= d
        assert base + c["offset"] + 2 * n <= malloc_bytes, c
    with open(OUT, "w") as f:
        json.dump(layout, f, indent=0, separators=(",", ":"))
    print("%d qinfo records, %d constant tensors -> %s" % (len(qinfo), len(consts), OUT))


if __name__ == "__main__":
    main()
