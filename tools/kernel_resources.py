"""Register / scratch / LDS figures of every kernel in the shipped library, read from the code objects inside the .so
(the .hip_fatbin section holds one clang offload bundle per translation unit; each bundle carries a gfx950 ELF whose
NT_AMDGPU_METADATA note lists the kernels):
    python tools/kernel_resources.py [libmzk_hip.so] [substring ...]
As a module: kernel_resources(path) -> {kernel name: {vgpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds, max_flat_wg}}."""
import os, re, struct, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _fatbin(so):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "fatbin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + out, so, os.path.join(d, "discard")])
        return open(out, "rb").read()


def code_objects(so):
    """the gfx950 ELF images of every bundle in the library"""
    blob = _fatbin(so)
    pos, out = 0, []
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        (n,) = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        at = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, at)
            triple = blob[at + 24:at + 24 + tlen].decode()
            at += 24 + tlen
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos += len(MAGIC)
    return out


def kernel_resources(so):
    res = {}
    for img in code_objects(so):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img); f.flush()
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
            def g(key, blk=blk):
                m = re.search(r"\.%s:\s*(\S+)" % key, blk)
                return m.group(1) if m else None
            name = g("name")
            if not name:
                continue
            for filt in (os.path.join(LLVM, "llvm-cxxfilt"), "c++filt"):      # (this image ships binutils' c++filt, not LLVM's)
                try:
                    name = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() or name
                    break
                except Exception:
                    pass
            res[name] = {"vgpr": int(g("vgpr_count") or 0), "sgpr": int(g("sgpr_count") or 0), "vgpr_spill": int(g("vgpr_spill_count") or 0),
                         "sgpr_spill": int(g("sgpr_spill_count") or 0), "scratch": int(g("private_segment_fixed_size") or 0),
                         "lds": int(g("group_segment_fixed_size") or 0), "max_flat_wg": int(g("max_flat_workgroup_size") or 0)}
    return res


# the kernels bench.py prices (substrings of the demangled names): tests/test_abi_load.py guards their spills and scratch, and
# `python tools/kernel_resources.py --priced` prints their table (profiles/r05final_kernel_resources.txt)
PRICED = ("k_seg_accumulate", "k_seg_combine", "k_ntt_strided", "k_ntt_last", "k_direct_accumulate", "k_direct_finish", "k_many_sort1", "k_many_count",
          "k_many_scatter", "k_open_many", "k_fine_scatter", "k_fine_count", "k_coarse_scatter", "k_coarse_count", "k_merkle", "k_fri_fold", "k_reduce_tail_row",
          "k_prepare_points", "k_window_combine_row")


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = sys.argv[1:]
    so = args.pop(0) if args and args[0].endswith(".so") else os.path.join(here, "myzkp_amd", "libmzk_hip.so")
    if "--priced" in args:
        args = [a for a in args if a != "--priced"] + list(PRICED)
    r = kernel_resources(so)
    print("%-110s %5s %5s %7s %7s %8s %7s" % ("kernel", "vgpr", "sgpr", "v-spill", "s-spill", "scratch", "lds"))
    for k in sorted(r):
        short = k.replace("mzk::", "").replace("void ", "")
        if args and not any(a in short for a in args):
            continue
        v = r[k]
        print("%-110s %5d %5d %7d %7d %8d %7d" % (short[:110], v["vgpr"], v["sgpr"], v["vgpr_spill"], v["sgpr_spill"], v["scratch"], v["lds"]))
