"""The hand-off of the bias-split forward dynamics (mecano_amd/csrc/mh_zv_kernels.h: zv_bias_group -> zv_wait / zv_aba_group), checked on
the MACHINE CODE of a topology-specialised code object.

The C++ source expresses the protocol with relaxed agent-scope atomics (which hipcc lowers to sc1 accesses), an inline `s_waitcnt
vmcnt(0)` and workgroup barriers -- the form /opt/skills/guides/MI355X_MICROARCH.md lists as valid without agent-scope fences.  Nothing in
the language keeps a future compiler from re-scoping one access or moving the flag store, so tests/test_handoff_isa.py disassembles every
registered code object and checks the instruction stream itself:

  producer   every published row store (global_store_dwordx2 with a scope bit) is followed by `s_waitcnt vmcnt(0)`, then `s_barrier`, and
             only then by the flag store (global_store_dword) with the SAME scope bits; no flag store sits between the rows and that barrier;
  consumer   the flag is polled with `global_load_dword ... sc1` (a first look and a loop with s_sleep); a workgroup barrier separates the
             last poll from the first row load; every row load behind it is `global_load_dwordx2 ... sc1`;
  give-up    the wall-clock limit's error word is stored at system scope (sc0 sc1).

usage: python tools/isa_handoff.py <libmecano_hip_topo_*.so | file.s> ...
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(so_path: str) -> str:
    """Device code (gfx950) of a HIP shared object as llvm-objdump text."""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path, os.path.join(tmp, "copy.so")])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co], text=True)


def kernels(text: str) -> dict:
    """{mangled name: [(mnemonic, operands)]} from llvm-objdump output or from a compiler listing (-S)."""
    out, cur = {}, None
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\w+)>:\s*$", line) or re.match(r"^(_Z\w+):", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        body = line.split("//")[0].split(";")[0].strip()
        if not body or body.startswith(".") or body.endswith(":"):
            continue
        parts = body.split(None, 1)
        if re.match(r"^(s_|v_|ds_|global_|buffer_|flat_|scratch_)", parts[0]):
            cur.append((parts[0], parts[1] if len(parts) > 1 else ""))
    return out


def scope(operands: str) -> str:
    return " ".join(b for b in ("sc0", "sc1") if re.search(r"\b" + b + r"\b", operands))


def check_handoff(instrs) -> list:
    """Violations of the hand-off protocol in one spec_zv_kernel instruction stream (empty list: the stream is as the protocol needs it)."""
    bad = []
    n = len(instrs)
    idx = lambda pred: [i for i, (op, a) in enumerate(instrs) if pred(op, a)]
    barriers = idx(lambda op, a: op == "s_barrier")
    drains = idx(lambda op, a: op == "s_waitcnt" and re.search(r"vmcnt\(0\)", a))
    flag_stores = idx(lambda op, a: op == "global_store_dword")
    row_stores = idx(lambda op, a: op == "global_store_dwordx2" and scope(a) in ("sc0", "sc1"))
    row_loads = idx(lambda op, a: op == "global_load_dwordx2" and scope(a) == "sc1")
    # (the bias job reads its mailbox word the same way, further down the kernel: the flag polls are the ones in front of the row loads)
    polls = [i for i in idx(lambda op, a: op == "global_load_dword" and scope(a) == "sc1") if not row_loads or i < row_loads[0]]
    # ---- producer
    if not [i for i in row_stores if scope(instrs[i][1]) == "sc1"]:
        bad.append("producer: no write-through (sc1) row store found")
    for i in row_stores:
        sc = scope(instrs[i][1])
        nb = next((b for b in barriers if b > i), None)
        if nb is None:
            bad.append(f"producer: row store #{i} ({sc}) is not followed by a workgroup barrier")
            continue
        if not [d for d in drains if i < d < nb]:
            bad.append(f"producer: no `s_waitcnt vmcnt(0)` between row store #{i} and the barrier #{nb} in front of the flag store")
        early = [f for f in flag_stores if i < f < nb]
        if early:
            bad.append(f"producer: flag-sized store #{early[0]} sits between row store #{i} and the barrier #{nb}: the flag can overtake its rows")
        nf = next((f for f in flag_stores if f > nb and scope(instrs[f][1]) == sc), None)
        if nf is None or nf - nb > 200:
            bad.append(f"producer: no {sc} flag store within 200 instructions behind barrier #{nb}")
    # ---- consumer
    if len(polls) < 2:
        bad.append(f"consumer: expected a first look and a polling loop on the flag (global_load_dword sc1), found {len(polls)}")
    elif not [i for i, (op, a) in enumerate(instrs) if op == "s_sleep" and polls[0] < i < polls[-1] + 40]:
        bad.append("consumer: the polling loop has no s_sleep")
    if not row_loads:
        bad.append("consumer: no sc1 row load (global_load_dwordx2 sc1) found")
    elif polls:
        first = row_loads[0]
        fence = [b for b in barriers if polls[-1] < b < first]
        if not fence:
            bad.append(f"consumer: no workgroup barrier between the last poll #{polls[-1]} and the first row load #{first}")
        else:
            nb = next((b for b in barriers if b > row_loads[-1]), n)
            plain = [i for i, (op, a) in enumerate(instrs) if fence[-1] < i < nb and op == "global_load_dwordx2" and scope(a) != "sc1"]
            if plain:
                bad.append(f"consumer: row load #{plain[0]} behind the barrier is not sc1 (it may be served from this CU's L1)")
    # ---- give-up
    if not [f for f in flag_stores if scope(instrs[f][1]) == "sc0 sc1"]:
        bad.append("give-up: the error word is not stored at system scope (sc0 sc1)")
    return bad


def zv_kernels(text: str) -> dict:
    return {k: v for k, v in kernels(text).items() if "spec_zv_kernel" in k}


if __name__ == "__main__":
    rc = 0
    for path in sys.argv[1:]:
        text = open(path).read() if path.endswith(".s") else disassemble(path)
        found = zv_kernels(text)
        if not found:
            print(f"{path}: no bias-split kernel in this code object")
        for name, instrs in found.items():
            bad = check_handoff(instrs)
            print(f"{path}: {name[:60]}: {len(instrs)} instructions, " + ("hand-off as required" if not bad else "; ".join(bad)))
            rc |= 1 if bad else 0
    sys.exit(rc)
