"""Fingerprint of the kernel sources a recorded counter profile belongs to.

A profile under profiles/ (HBM traffic, SQ counters) is only as good as the kernel it was taken from: bench.py quotes such
records (`roofline.traffic`) and must be able to say when the kernel has changed since.  File dates do not survive a git
checkout or a gpurun snapshot, so the summaries carry `# source_fingerprint msm=<hex> ntt=<hex>` in their first line and
bench.py compares that with the tree it runs from (`traffic_stale`).

    python tools/source_fingerprint.py            -> the header line for the current tree
"""
import hashlib, os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "myzkp_amd", "csrc")
FAMILIES = {
    # what the accumulate / sort kernels of the MSM are compiled from
    "msm": ["mzk_msm.hip", "mzk_ec.h", "mzk_field.h", "mzk_field_asm.h", "mzk_coop.h", "mzk_glv.h", "mzk_common.h"],
    # the transform passes
    "ntt": ["mzk_ntt.hip", "mzk_field.h", "mzk_field_asm.h", "mzk_common.h"],
}


def fingerprint(family):
    h = hashlib.sha256()
    for name in FAMILIES[family]:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def header_line():
    return "# source_fingerprint " + " ".join("%s=%s" % (k, fingerprint(k)) for k in sorted(FAMILIES))


def parse_header(path):
    """{family: hex} from the first lines of a recorded summary ({} if it carries none: a record from before round 6)."""
    try:
        with open(path) as f:
            for _ in range(3):
                line = f.readline()
                if line.startswith("# source_fingerprint"):
                    return dict(tok.split("=", 1) for tok in line.split()[2:] if "=" in tok)
    except OSError:
        pass
    return {}


if __name__ == "__main__":
    print(header_line())
