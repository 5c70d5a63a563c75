"""Registers, scratch and static LDS of the shipped kernels (llvm-readelf --notes of the code objects inside the .so files):
python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import subprocess, re, os, sys, tempfile
L = "/opt/rocm/lib/llvm/bin"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("spec_zv_kernel", "spec_zvf_kernel", "spec_zvb", "aba_dfs_kernel<float", "rnea_dfs_kernel<float", "rows_to_columns", "columns_to_rows", "spec_crba", "spec_split")
print("# llvm-readelf --notes of the shipped binaries: registers, scratch and static LDS of the kernels DESIGN.md quotes")
print("# file | kernel | VGPRs | AGPRs | SGPRs | scratch bytes per lane | static LDS bytes | VGPR spills")
with tempfile.TemporaryDirectory() as tmp:
    for so in ("mecano_amd/libmecano_hip_topo_b5c1e26c784c54fa.so", "mecano_amd/libmecano_hip.so"):
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([L + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, os.path.join(root, so), os.path.join(tmp, "copy.so")])
        blob, magic = open(fat, "rb").read(), b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [i for i in range(len(blob)) if blob.startswith(magic, i)]
        for n, a in enumerate(starts):
            part, co = os.path.join(tmp, "b.bin"), os.path.join(tmp, "dev.co")
            open(part, "wb").write(blob[a:starts[n + 1] if n + 1 < len(starts) else len(blob)])
            subprocess.check_call([L + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
            notes = subprocess.check_output([L + "/llvm-readelf", "--notes", co], text=True)
            for blk in notes.split("- .agpr_count:")[1:]:
                g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
                dem = subprocess.check_output(["c++filt", g("name")], text=True).strip()
                if any(k in dem for k in KEYS):
                    print(" | ".join((os.path.basename(so), dem[:120], g("vgpr_count"), blk.split("\n")[0].strip(), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("vgpr_spill_count"))))
