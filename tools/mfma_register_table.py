"""Register / scratch table of every matrix-pipe kernel variant, from the reports the build leaves beside the objects
(pybold_amd/csrc/build/mfma*.res: `-Rpass-analysis=kernel-resource-usage` of the very compile that made the object).
DESIGN.md's table is this script's output, so it cannot go stale:

    python tools/mfma_register_table.py            # markdown to stdout
    python tools/mfma_register_table.py --check    # exit 1 if any variant uses scratch (tests/test_isa_lint.py)
"""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def variants():
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "pybold_amd", "csrc", "build", "mfma*.res"))):
        for b in open(f).read().split("Function Name: ")[1:]:
            name = b.split()[0]

            def num(key):
                return int(re.search(key + r": (\d+)", b).group(1))
            m1 = re.search(r"fista_mfma_kernelILi(\d+)ELb(\d)ELb(\d)ELb(\d)ELi(\d)ELb(\d)", name)
            m2 = re.search(r"fista_mfma2_kernelILi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELi(\d)", name)
            m4 = re.search(r"fista_mfma4_kernelILi(\d+)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELi(\d)", name)
            if m1:
                nb, j, dev, cert, nt, loops = (int(x) for x in m1.groups())
                kind = "fista_mfma_kernel<%d>" % nb
                var = "+".join([v for v, on in (("cost trace", j and not cert), ("certificate", cert), ("taps from device", dev),
                                                ("3 near tiles", nt == 3), ("_loops_deconv rule", loops)) if on]) or "plain"
            elif m2:
                a, b2, dev, j, cert, loops, nt = (int(x) for x in m2.groups())
                kind = "fista_mfma2_kernel<%d,%d>" % (a, b2)
                var = "+".join([v for v, on in (("cost trace", j and not cert), ("certificate", cert), ("taps from device", dev),
                                                ("_loops_deconv rule", loops), ("3 near tiles", nt == 3)) if on]) or "plain"
            elif m4:
                a, dev, j, cert, loops, nt = (int(x) for x in m4.groups())
                kind = "fista_mfma4_kernel<%d>" % a
                var = "+".join([v for v, on in (("cost trace", j and not cert), ("certificate", cert), ("taps from device", dev),
                                                ("_loops_deconv rule", loops), ("3 near tiles", nt == 3)) if on]) or "plain"
            else:
                continue
            out.append(dict(kernel=kind, variant=var, vgpr=num("VGPRs"), agpr=num("AGPRs"), scratch=num(r"ScratchSize \[bytes/lane\]"),
                            lds=num(r"LDS Size \[bytes/block\]"), occ=num(r"Occupancy \[waves/SIMD\]"), file=os.path.basename(f)))
    return out


# Explicit allowances (bytes of scratch per lane), each with its measured reason.  fista_mfma4_kernel<9> certificate + three near
# tiles: the lane index (one register) is parked in scratch at kernel entry and read back once per role prologue and twice per
# iteration next to a barrier; A = 8 and A = 10 fit (226 accumulator registers).  Nothing under a partial exec mask
# (tools/isa_spill_lint.py on build/mfma4_9.s).
# fista_mfma2_kernel<10,10> _loops_deconv rule + three near tiles: one register stored in a role's prologue, read back once
# before its loop (build/mfma2_10_10.s: no scratch instruction between the loop's barriers).
ALLOW = {("fista_mfma4_kernel<9>", "certificate+3 near tiles"): 8,
         ("fista_mfma2_kernel<10,10>", "_loops_deconv rule+3 near tiles"): 8}


def main():
    vs = variants()
    if "--check" in sys.argv:
        bad = [v for v in vs if v["scratch"] > ALLOW.get((v["kernel"], v["variant"]), 0)]
        for v in bad:
            print("%(kernel)s %(variant)s: %(scratch)d B of scratch per lane" % v)
        print("%d variants, %d with scratch" % (len(vs), len(bad)))
        sys.exit(1 if bad or not vs else 0)
    print("| kernel | variant | VGPRs | AGPRs | scratch (B/lane) | waves/SIMD |")
    print("|---|---|---|---|---|---|")
    for v in vs:
        print("| `%(kernel)s` | %(variant)s | %(vgpr)d | %(agpr)d | %(scratch)d | %(occ)d |" % v)


if __name__ == "__main__":
    main()
