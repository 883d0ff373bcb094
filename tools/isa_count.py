"""Instruction mix of the main loop of every kernel in a hipcc -S listing (development aid).
usage: python tools/isa_count.py build/fast_19_30.s [substring of the mangled name]"""
import collections
import re
import sys


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(line.rstrip())
            if line.strip().startswith("s_endpgm"):
                yield name, body
                name = None


def main_loop(body):
    """Largest span between a label and a later backward branch to it."""
    labels = {}
    best = (0, 0)
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels:
            lo = labels[m.group(1)]
            if i - lo > best[1] - best[0]:
                best = (lo, i)
    return body[best[0]:best[1] + 1]


def classify(op):
    if op.startswith("v_pk_fma_f32"): return "v_pk_fma_f32"
    if op.startswith("v_pk_"): return "v_pk_other"
    if "dpp" in op: return "dpp"
    if op.endswith("_f64") or "_f64_" in op: return "f64"
    if op.startswith("v_cvt"): return "cvt"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith("v_mov"): return "v_mov"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "salu"
    return "other"


for name, body in kernels(sys.argv[1]):
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    loop = main_loop(body)
    c = collections.Counter()
    for l in loop:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if "dpp" in t and op.startswith("v_"):
            op = op + "_dpp"
        c[classify(op)] += 1
    valu = sum(v for k, v in c.items() if k not in ("lds", "vmem", "s_waitcnt", "s_nop", "salu", "other"))
    print(name)
    print("   loop lines %d  VALU %d  |" % (sum(c.values()), valu), dict(sorted(c.items())))
