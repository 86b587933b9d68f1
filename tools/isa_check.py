"""Static checks on the gfx950 code objects of the library (no GPU needed): the hazard rules of DESIGN.md section 5
turned into something a build can fail on.  Used by tests/test_build_cpu.py; `python tools/isa_check.py` prints a report.

The checks work on `llvm-objdump -d` text of the device code object embedded in each build/*.o:

* pending_lds_reads_at_barriers  -- a forward may-analysis over the kernel's control-flow graph: an LDS read (`ds_read*`)
  is "pending" until an `s_waitcnt` with `lgkmcnt(0)` retires it.  A kernel that stages operands by LDS-DMA
  (`buffer_load ... lds`) re-fills ring slots right behind its barriers, so a fragment read still in flight AT a barrier
  races with the DMA that overwrites its slot (the round-2 pw_stream write-after-read race).
* ds_read_b128_under_partial_exec -- 128-bit LDS reads executed while EXEC may be partial (between an
  `s_and_saveexec` / `v_cmpx` / `s_andn2 exec` and the `s_or_b64 exec, exec, ...` that restores it): the access shape that
  returned wrong data next to another kernel's MFMA waves (rule (i)).
* lds_dma_kernels / has_instr -- helpers for the per-object rules (no ds_read_b128 at all in the pre / post kernels).
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "rs-face-detection_amd", "build")

_INS = re.compile(r"^\t(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_FUNC = re.compile(r"^([0-9a-f]+) <(\S+)>:")
_TGT = re.compile(r"<(\S+?)\+0x([0-9a-f]+)>\s*$")


def extract_code_object(obj, tmp):
    base = os.path.join(tmp, os.path.basename(obj))
    fat, co = base + ".fat", base + ".co"
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    return co


def disassemble(obj, tmp=None):
    """-> {mangled kernel name: [(addr, mnemonic, operands, branch_target_addr or None), ...]}"""
    own = tmp is None
    if own:
        tmpd = tempfile.TemporaryDirectory()
        tmp = tmpd.name
    co = extract_code_object(obj, tmp)
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    kernels, cur, base = {}, None, {}
    for line in txt.splitlines():
        m = _FUNC.match(line)
        if m:
            cur = m.group(2)
            base[cur] = int(m.group(1), 16)
            kernels[cur] = []
            continue
        m = _INS.match(line)
        if not m or cur is None:
            continue
        mnem, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
        tgt = None
        if mnem.startswith("s_cbranch") or mnem == "s_branch":
            t = _TGT.search(line)
            if t:
                tgt = base.get(t.group(1), base[cur]) + int(t.group(2), 16)
        kernels[cur].append((addr, mnem, ops, tgt))
    if own:
        tmpd.cleanup()
    return kernels


def has_instr(ins, prefix):
    return [hex(a) for a, m, o, t in ins if m.startswith(prefix)]


def uses_lds_dma(ins):
    return any(m.startswith("buffer_load") and re.search(r"\blds\b", o) for a, m, o, t in ins)


def _blocks(ins):
    """basic blocks as (start index, end index exclusive) + successor lists"""
    addr_idx = {a: i for i, (a, m, o, t) in enumerate(ins)}
    leaders = {0}
    for i, (a, m, o, t) in enumerate(ins):
        if t is not None:
            if t in addr_idx:
                leaders.add(addr_idx[t])
            if i + 1 < len(ins):
                leaders.add(i + 1)
        if m == "s_endpgm" and i + 1 < len(ins):
            leaders.add(i + 1)
    starts = sorted(leaders)
    blocks, of_start = [], {}
    for k, s in enumerate(starts):
        e = starts[k + 1] if k + 1 < len(starts) else len(ins)
        of_start[s] = k
        blocks.append((s, e))
    succ = []
    for s, e in blocks:
        a, m, o, t = ins[e - 1]
        out = []
        if m == "s_endpgm":
            pass
        elif m == "s_branch":
            if t in addr_idx:
                out.append(of_start[addr_idx[t]])
        else:
            if t is not None and t in addr_idx:
                out.append(of_start[addr_idx[t]])
            if e < len(ins):
                out.append(of_start[e])
        succ.append(out)
    return blocks, succ


def _dataflow(ins, transfer, init, join):
    """forward fixpoint; transfer(state, instr, report) -> state; returns the state ENTERING every instruction"""
    blocks, succ = _blocks(ins)
    entry = [None] * len(blocks)
    entry[0] = init
    work = [0]
    while work:
        b = work.pop()
        st = entry[b]
        s, e = blocks[b]
        for i in range(s, e):
            st = transfer(st, ins[i], None)
        for n in succ[b]:
            new = st if entry[n] is None else join(entry[n], st)
            if new != entry[n]:
                entry[n] = new
                work.append(n)
    return blocks, entry


def _lgkm0(ops):
    return re.search(r"lgkmcnt\(0\)", ops) is not None


def pending_lds_reads_at_barriers(ins):
    """addresses of s_barrier instructions that an LDS read issued earlier may still be in flight at"""
    def tr(st, instr, _):
        a, m, o, t = instr
        if m.startswith("ds_read") or m.startswith("ds_load"):
            return True
        if m == "s_waitcnt" and _lgkm0(o):
            return False
        return st
    blocks, entry = _dataflow(ins, tr, False, lambda x, y: x or y)
    bad = []
    for (s, e), st in zip(blocks, entry):
        if st is None:
            continue
        for i in range(s, e):
            if ins[i][1] == "s_barrier" and st:
                bad.append(hex(ins[i][0]))
            st = tr(st, ins[i], None)
    return bad


_SAVEEXEC = re.compile(r"^s_(and|andn2|or|xor|andn1|orn1|orn2|nand|nor|xnor)_saveexec_b64$")
_SREG = re.compile(r"^s\[(\d+):(\d+)\]$|^s(\d+)$")


def _sregs(op):
    m = _SREG.match(op)
    if not m:
        return ()
    if m.group(3) is not None:
        return (int(m.group(3)),)
    return tuple(range(int(m.group(1)), int(m.group(2)) + 1))


def ds_read_b128_under_partial_exec(ins):
    """addresses of ds_read_b128 that execute at a point where EXEC may be partial.

    State: (exec_may_be_partial, {sgpr pair holding a saved EXEC: was that EXEC possibly partial}).  A wave starts with EXEC
    all ones (every launch uses whole waves); `s_*_saveexec sX` remembers the state in sX and narrows; `s_or_b64 exec, exec, sX`
    / `s_mov_b64 exec, sX` restore the remembered state (exec is a subset of what sX saved); any other write to EXEC, and
    `v_cmpx`, narrow it; overwriting sX forgets it (restoring from an unknown register counts as partial)."""
    def tr(st, instr, _):
        part, saved = st
        a, m, o, t = instr
        ops = [x.strip() for x in o.split(",")] if o else []
        dst = ops[0] if ops else ""
        if _SAVEEXEC.match(m):
            saved = dict(saved)
            for k in [k for k in saved if set(k) & set(_sregs(dst))]:
                del saved[k]
            saved[_sregs(dst)] = part
            return (True, tuple(sorted(saved.items())) and saved)
        if m.startswith("v_cmpx"):
            return (True, saved)
        if dst == "exec":
            if m == "s_or_b64" and len(ops) == 3 and "exec" in ops[1:]:
                src = ops[2] if ops[1] == "exec" else ops[1]
                return (saved.get(_sregs(src), True), saved)
            if m == "s_mov_b64":
                return (False, saved) if ops[1] == "-1" else (saved.get(_sregs(ops[1]), True), saved)
            return (True, saved)
        regs = set(_sregs(dst))
        if regs and m.startswith(("s_", "v_cmp", "v_readlane", "v_readfirstlane")) and not m.startswith(("s_cbranch", "s_cmp", "s_waitcnt", "s_bitcmp")):
            hit = [k for k in saved if set(k) & regs]
            if hit:
                saved = {k: v for k, v in saved.items() if k not in hit}
        return (part, saved)

    def norm(st):
        return (st[0], tuple(sorted(st[1].items())))

    def join(x, y):
        (px, sx), (py, sy) = x, y
        sx, sy = dict(sx), dict(sy)
        out = {k: (sx[k] or sy[k]) for k in sx if k in sy}
        return (px or py, tuple(sorted(out.items())))

    def tr_n(st, instr, r):
        part, saved = st
        return norm(tr((part, dict(saved)), instr, r))

    blocks, entry = _dataflow(ins, tr_n, (False, ()), join)
    bad = []
    for (s, e), st in zip(blocks, entry):
        if st is None:
            continue
        for i in range(s, e):
            if ins[i][1] == "ds_read_b128" and st[0]:
                bad.append(hex(ins[i][0]))
            st = tr_n(st, ins[i], None)
    return bad


_VMCNT = re.compile(r"vmcnt\((\d+)\)")


def _is_store(m):
    return m.startswith(("buffer_store", "global_store", "flat_store", "scratch_store", "buffer_atomic", "global_atomic", "flat_atomic"))


def _publishes(ins, i, addr_idx, depth=24):
    """does the code right behind instruction i hand data to other waves (an LDS add / write or a barrier) before the next
    wait on the vector-memory counter?  Follows fall-through and branch targets for a few instructions."""
    seen, work = set(), [(i + 1, depth)]
    while work:
        j, d = work.pop()
        while j < len(ins) and d > 0 and j not in seen:
            seen.add(j)
            a, m, o, t = ins[j]
            if m == "s_barrier" or m.startswith(("ds_add", "ds_write")):
                return True
            if m == "s_waitcnt" and _VMCNT.search(o):
                break
            if m == "s_endpgm":
                break
            if t is not None and t in addr_idx:
                work.append((addr_idx[t], d - 1))
                if m == "s_branch":
                    break
            j += 1
            d -= 1
    return False


def counted_vmcnt_with_store_in_flight(ins):
    """addresses of PUBLISHING counted waits -- `s_waitcnt vmcnt(N > 0)` followed by an LDS add / write or a barrier, i.e. the
    wait that tells other waves "this LDS-DMA tile has landed" -- that may execute while a vector store (or atomic) of the same
    wave is outstanding.

    "All but my N youngest operations are done" publishes the right tile only if the N youngest are the later DMAs.  The
    persistent kernels of round 2 had output stores queued among them and saw stale tiles; they drain (vmcnt(0)) ever since.
    The ring kernels (kernels_ring.hip) keep counted waits, in loader waves that never store -- this check holds them to it.
    Forward may-analysis: state = (dma pending, store pending); an s_waitcnt with vmcnt(0) clears both.  Waits hipcc places in
    front of an epilogue's register uses are not publishing waits and are left alone."""
    addr_idx = {a: i for i, (a, m, o, t) in enumerate(ins)}

    def tr(st, instr, _):
        dma, store = st
        a, m, o, t = instr
        if m.startswith("buffer_load") and re.search(r"\blds\b", o):
            dma = True
        elif _is_store(m):
            store = True
        elif m == "s_waitcnt":
            v = _VMCNT.search(o)
            if v and int(v.group(1)) == 0:
                dma, store = False, False
        return (dma, store)
    blocks, entry = _dataflow(ins, tr, (False, False), lambda x, y: (x[0] or y[0], x[1] or y[1]))
    bad = []
    for (s, e), st in zip(blocks, entry):
        if st is None:
            continue
        for i in range(s, e):
            a, m, o, t = ins[i]
            if m == "s_waitcnt":
                v = _VMCNT.search(o)
                if v and int(v.group(1)) > 0 and st[0] and st[1] and _publishes(ins, i, addr_idx):
                    bad.append(hex(a))
            st = tr(st, ins[i], None)
    return bad


def counted_vmcnt_waits(ins):
    """addresses of every PUBLISHING `s_waitcnt vmcnt(N > 0)` reached with an LDS-DMA possibly outstanding (the waits the rule is about)"""
    def tr(st, instr, _):
        a, m, o, t = instr
        if m.startswith("buffer_load") and re.search(r"\blds\b", o):
            return True
        if m == "s_waitcnt":
            v = _VMCNT.search(o)
            if v and int(v.group(1)) == 0:
                return False
        return st
    addr_idx = {a: i for i, (a, m, o, t) in enumerate(ins)}
    blocks, entry = _dataflow(ins, tr, False, lambda x, y: x or y)
    out = []
    for (s, e), st in zip(blocks, entry):
        if st is None:
            continue
        for i in range(s, e):
            a, m, o, t = ins[i]
            if m == "s_waitcnt" and st:
                v = _VMCNT.search(o)
                if v and int(v.group(1)) > 0 and _publishes(ins, i, addr_idx):
                    out.append(hex(a))
            st = tr(st, ins[i], None)
    return out


def report(objs=("kernels_pre", "kernels_post", "kernels_conv", "kernels_ring", "kernels_f32")):
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for f in objs:
            ks = disassemble(os.path.join(BUILD, f + ".o"), tmp)
            for name, ins in ks.items():
                rows.append((f, name, len(ins), uses_lds_dma(ins), len(has_instr(ins, "ds_read_b128")),
                             pending_lds_reads_at_barriers(ins), ds_read_b128_under_partial_exec(ins),
                             counted_vmcnt_with_store_in_flight(ins), counted_vmcnt_waits(ins)))
    return rows


if __name__ == "__main__":
    for f, name, n, dma, nb128, bar, ex, cnt_bad, cnt in report():
        flag = ("  BARRIER-WITH-PENDING-LDS-READ " + ",".join(bar) if dma and bar else "") + ("  B128-UNDER-PARTIAL-EXEC " + ",".join(ex) if ex else "") + \
               ("  COUNTED-VMCNT-WITH-STORE-IN-FLIGHT " + ",".join(cnt_bad) if cnt_bad else "")
        print("%-13s %-90s %6d instr  lds-dma=%d  ds_read_b128=%d  counted-vmcnt-behind-dma=%d%s" % (f, name[:90], n, dma, nb128, len(cnt), flag))
    sys.exit(0)
