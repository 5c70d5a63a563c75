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

Code objects with identity index maps run the hand-off in two stages (limb columns under flag A, trunk columns under flag B: see
check_two_stage); the stream tells which form it is.  Round 5: stage one needs no flag -- the limb columns signal themselves (the matrix
holds a signalling-NaN sentinel between launches, the consumer polls the columns): see check_self_signal.

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
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path, os.path.join(tmp, "copy.so")])
        # a code object linked from several translation units carries one offload bundle per unit, one behind the other
        blob, magic, text = open(fat, "rb").read(), b"__CLANG_OFFLOAD_BUNDLE__", ""
        starts = [i for i in range(len(blob)) if blob.startswith(magic, i)]
        for n, a in enumerate(starts):
            part, co = os.path.join(tmp, f"bundle{n}.bin"), os.path.join(tmp, f"dev{n}.co")
            with open(part, "wb") as f:
                f.write(blob[a:starts[n + 1] if n + 1 < len(starts) else len(blob)])
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
            text += subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co], text=True)
        return text


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


def is_two_stage(instrs) -> bool:
    """The two-stage form counts its publishing waves in through an LDS atomic right in front of a flag store."""
    for i, (op, a) in enumerate(instrs):
        if op == "global_store_dword" and scope(a) == "sc1" and any(o.startswith("ds_add_rtn") for o, _ in instrs[max(0, i - 40):i]):
            return True
    return False


def check_two_stage(instrs) -> list:
    """The two-stage hand-off (zv_bias_group2 / zv_aba_group2): flag A covers the limb columns of waves 1-3 (each drains its stores, counts
    itself in through LDS, the last arrival stores the flag), flag B the trunk columns of wave 0 (drains, stores the flag itself); every
    consumer wave polls for itself and loads only behind its own poll."""
    bad = []
    n = len(instrs)
    idx = lambda pred: [i for i, (op, a) in enumerate(instrs) if pred(op, a)]
    drains = idx(lambda op, a: op == "s_waitcnt" and re.search(r"vmcnt\(0\)", a))
    small_stores = idx(lambda op, a: op == "global_store_dword")
    flag_stores = [i for i in small_stores if scope(instrs[i][1]) == "sc1"]
    row_stores = idx(lambda op, a: op == "global_store_dwordx2" and scope(a) != "")
    row_loads = idx(lambda op, a: op == "global_load_dwordx2" and scope(a) == "sc1")
    polls = idx(lambda op, a: op == "global_load_dword" and scope(a) == "sc1")
    # ---- producer
    if not row_stores:
        bad.append("producer: no scoped column store (global_store_dwordx2 sc1) found")
    for i in row_stores:
        if scope(instrs[i][1]) != "sc1":
            bad.append(f"producer: column store #{i} is {scope(instrs[i][1])}, not write-through (sc1)")
        nd = next((d for d in drains if d > i), None)
        if nd is None:
            bad.append(f"producer: column store #{i} is not followed by `s_waitcnt vmcnt(0)`")
            continue
        early = [f for f in small_stores if i < f < nd]
        if early:
            bad.append(f"producer: flag-sized store #{early[0]} sits between column store #{i} and its wave's drain #{nd}: the flag can overtake its columns")
        nf = next((f for f in flag_stores if f > nd), None)
        if nf is None or nf - nd > 120:
            bad.append(f"producer: no sc1 flag store within 120 instructions behind the drain #{nd} of column store #{i}")
    counted = [f for f in flag_stores if any(o.startswith("ds_add_rtn") for o, _ in instrs[max(0, f - 40):f])]
    if not counted:
        bad.append("producer: no flag store behind an LDS arrival counter (the limb columns are published by three waves)")
    for f in counted:
        add = max(i for i in range(max(0, f - 40), f) if instrs[i][0].startswith("ds_add_rtn"))
        if not [d for d in drains if row_stores and max((r for r in row_stores if r < add), default=-1) < d < add]:
            bad.append(f"producer: the arrival count #{add} in front of flag store #{f} is not behind its wave's `s_waitcnt vmcnt(0)`")
    # ---- consumer
    if len(polls) < 4:
        bad.append(f"consumer: expected a first look and a polling loop on each of the two flags (global_load_dword sc1), found {len(polls)}")
    if not [i for i, (op, a) in enumerate(instrs) if op == "s_sleep" and polls and polls[0] < i < polls[-1] + 40]:
        bad.append("consumer: the polling loops have no s_sleep")
    if not row_loads:
        bad.append("consumer: no sc1 column load (global_load_dwordx2 sc1) found")
    elif polls:
        if row_loads[0] < polls[0]:
            bad.append(f"consumer: column load #{row_loads[0]} comes before the first poll #{polls[0]}")
        # (the jobs' code lies one behind the other in the stream and shares one exit: the consumer's loads are those between its first
        # poll and its last sc1 column load -- a plain load among them would be a column read that may be served from this CU's L1)
        plain = [i for i, (op, a) in enumerate(instrs) if polls[0] < i < row_loads[-1] and op.startswith("global_load_dwordx") and scope(a) != "sc1"]
        if plain:
            bad.append(f"consumer: load #{plain[0]} between the first poll and the last column load is not sc1 (it may be served from this CU's L1)")
    # ---- give-up
    if not [f for f in small_stores if scope(instrs[f][1]) == "sc0 sc1"]:
        bad.append("give-up: the error word is not stored at system scope (sc0 sc1)")
    return bad


SENTINEL_HALF = "0x7ff4a5a5"  # mh_zv_kernels.h: ZV_SENTINEL, both 32-bit halves


def is_self_signal(instrs) -> bool:
    """The self-signalling form compares what it loads with the sentinel: its 32-bit half appears as a literal."""
    return any(SENTINEL_HALF in a for _, a in instrs)


def check_self_signal(instrs) -> list:
    """Two-stage hand-off whose FIRST stage has no flag (mh_zv_kernels.h, "Stage one without a flag"):
      producer   every published column store is write-through (global_store_dwordx2 sc1).  A run of column stores that is followed by its
                 wave's drain (`s_waitcnt vmcnt(0)` within 40 instructions) is the trunk's (stage two, flag B): no flag-sized store between the
                 run and the drain, an sc1 flag store within 120 instructions behind it; there is at least one such run.  The other runs are
                 the limbs' self-signalling columns (and, in the consumer, the sentinels written back): no rule beyond sc1 -- each element is
                 one aligned 8-byte store, valid by itself;
      consumer   the limb columns are POLLED: per wave a loop with s_sleep, sc1 column loads and a comparison with the sentinel, under a
                 wall-clock limit (s_memrealtime); flag B is polled with global_load_dword sc1 (a first look and a loop); every 8-byte
                 load from the first polled column to the last column load is sc1;
      give-up    error word and the poison word's host copy at system scope (sc0 sc1), the device poison word right behind them (sc1)."""
    bad = []
    idx = lambda pred: [i for i, (op, a) in enumerate(instrs) if pred(op, a)]
    drains = idx(lambda op, a: op == "s_waitcnt" and re.search(r"vmcnt\(0\)", a))
    small_stores = idx(lambda op, a: op == "global_store_dword")
    flag_stores = [i for i in small_stores if scope(instrs[i][1]) == "sc1"]
    col_stores = idx(lambda op, a: op == "global_store_dwordx2" and scope(a) != "")
    col_loads = idx(lambda op, a: op == "global_load_dwordx2" and scope(a) == "sc1")
    polls = idx(lambda op, a: op == "global_load_dword" and scope(a) == "sc1")
    sleeps = idx(lambda op, a: op == "s_sleep")
    sentinels = idx(lambda op, a: SENTINEL_HALF in a)
    # ---- producer
    if not col_stores:
        bad.append("producer: no scoped column store (global_store_dwordx2 sc1) found")
    for i in col_stores:
        if scope(instrs[i][1]) != "sc1":
            bad.append(f"producer: column store #{i} is {scope(instrs[i][1])}, not write-through (sc1)")
    runs = []  # [first, last] of consecutive column stores
    for i in col_stores:
        if runs and i - runs[-1][1] <= 3:
            runs[-1][1] = i
        else:
            runs.append([i, i])
    flagged = 0
    for first, last in runs:
        nd = next((d for d in drains if last < d <= last + 40), None)
        nf = next((f for f in flag_stores if f > last), None)
        if nd is None:
            if nf is not None and nf - last <= 40:
                bad.append(f"producer: flag-sized sc1 store #{nf} right behind the column stores #{first}..#{last} without a drain: the flag can overtake its columns")
            continue
        early = [f for f in small_stores if last < f < nd]
        if early:
            bad.append(f"producer: flag-sized store #{early[0]} sits between column store #{last} and its wave's drain #{nd}: the flag can overtake its columns")
            continue
        nf = next((f for f in flag_stores if f > nd), None)
        if nf is not None and nf - nd <= 120:
            flagged += 1
    if not flagged:
        bad.append("producer: no run of column stores that is drained and then flagged (stage two: the trunk's columns under flag B)")
    # ---- consumer
    loops = [s for s in sleeps if [l for l in col_loads if s < l <= s + 30] and [c for c in sentinels if s < c <= s + 60]
             and [i for i, (op, a) in enumerate(instrs) if s < i <= s + 90 and op == "s_memrealtime"]]
    if len(loops) < 4:
        bad.append(f"consumer: expected a polling loop (s_sleep, sc1 column loads, comparison with the sentinel, wall-clock limit) for each of the four waves, found {len(loops)}")
    if len(polls) < 3:
        bad.append(f"consumer: expected the poison word's load and a first look + polling loop on flag B (global_load_dword sc1), found {len(polls)}")
    if not col_loads:
        bad.append("consumer: no sc1 column load (global_load_dwordx2 sc1) found")
    else:
        plain = [i for i, (op, a) in enumerate(instrs) if col_loads[0] < i < col_loads[-1] and op.startswith("global_load_dwordx") and scope(a) != "sc1"]
        if plain:
            bad.append(f"consumer: load #{plain[0]} between the first and the last column load is not sc1 (it may be served from this CU's L1)")
    # ---- give-up
    system = [f for f in small_stores if scope(instrs[f][1]) == "sc0 sc1"]
    if not system:
        bad.append("give-up: the error word is not stored at system scope (sc0 sc1)")
    elif not [f for f in flag_stores if any(0 < f - s <= 4 for s in system)]:
        bad.append("give-up: no device poison word (global_store_dword sc1) stored right behind the error word")
    return bad


def check_handoff(instrs) -> list:
    """Violations of the hand-off protocol in one spec_zv_kernel instruction stream (empty list: the stream is as the protocol needs it)."""
    if is_self_signal(instrs):
        return check_self_signal(instrs)
    if is_two_stage(instrs):
        return check_two_stage(instrs)
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
