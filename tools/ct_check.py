#!/usr/bin/env python3
"""Constant-time check of the compiled kernels: no branch on, and no memory address from, secret scalars.

  python tools/ct_check.py [--kernel SUBSTR] [--verbose] [--json out.json]

The reference selects table entries by cmov scans and runs fixed-length loops (ge.rs:411-434, 488-500); this engine claims the same
property for its GPU kernels.  The claim is checked on the CODE THE GPU RUNS: each kernel unit is compiled to gfx950 assembly
(hipcc -S, no GPU needed) and a forward taint analysis runs over the control-flow graph of every kernel that handles secrets:

  sources   the vector loads whose address derives from a kernel argument declared secret below (the scalar arrays: private keys,
            nonces, DH secrets; for k_finish the projective results of a multiplication by one)
  flow      through every VALU / SALU / LDS / scratch instruction, register by register (a full overwrite with untainted inputs
            clears a register; LDS and scratch are one taint cell each), through VCC / SCC / EXEC / M0, to a fixed point over loops
  sinks     (1) the condition of every conditional branch (SCC, VCC, EXEC)        -> must be untainted
            (2) the address operands of every global / flat / scratch / LDS access and of every scalar load  -> must be untainted
            (3) EXEC at every memory access (which lanes touch memory)            -> must be untainted
            (4) indirect jumps, calls, register-indexed moves by a tainted M0      -> not allowed at all
  exempt    the LANE-SELECT operand of ds_bpermute_b32 / ds_permute_b32: the constant-time selection primitive of the fixed-base and
            small-batch kernels (it addresses a lane of the crossbar, not LDS memory; tools/microbench/bpermute_patterns.hip
            measures that its duration does not depend on the pattern).  The DATA an atomic carries may be tainted (k_mont_prep's
            "some scalar of this launch is not canonical" word); its address may not.

What the check does NOT cover, by construction: launch shapes chosen on the host from public properties of a whole batch
(ladder.skip_canonical: the word k_mont_prep collects decides between 252 and 256 ladder steps for the LAUNCH; kyb_mul_public_batch /
mul.short_scalars: declared-public multipliers) — include/kyber_ed25519.h, "timing" — and the hardware below the ISA.

Exit status 1 when any sink is tainted."""
import argparse
import concurrent.futures
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kyber-rs_amd", "csrc")

# kernel (mangled-name prefix) -> (unit, {kernarg byte offset of a pointer to secret data: what it is})
SECRET_ARGS = {
    "_Z12k_mul_ladderILi3E": ("kernels_ladder", {0: "scalars"}),
    "_Z12k_mul_ladderILi2E": ("kernels_ladder", {0: "scalars"}),
    "_Z17k_mul_ladder_pair": ("kernels_ladder", {0: "scalars"}),
    "_Z17k_mul_ladder_quad": ("kernels_ladder", {0: "scalars"}),
    "_Z19k_mul_ladder_pair_yILi1EE": ("kernels_ladder", {0: "scalars"}),
    "_Z19k_mul_ladder_pair_yILi2EE": ("kernels_ladder", {0: "scalars"}),      # (four lanes per item)
    "_Z23k_mul_ladder_pair_y_decILi1EE": ("kernels_ladder", {0: "scalars"}),      # the role of a workgroup depends on blockIdx only
    "_Z23k_mul_ladder_pair_y_decILi2EE": ("kernels_ladder", {0: "scalars"}),
    "_Z16k_ladder_recover": ("kernels_ladder", {0: "scalars", 24: "x-only state the ladder left (a function of the scalar)"}),
    "_Z11k_mont_prepPKim": ("kernels_ladder", {32: "scalars (top bits, canonical test)"}),
    "_Z12k_mul_base64ILb1ELi1024E": ("kernels_base", {0: "scalars", 8: "scalars_b"}),
    "_Z12k_mul_base64ILb1ELi768E": ("kernels_base", {0: "scalars", 8: "scalars_b"}),
    "_Z12k_mul_base64ILb1ELi256E": ("kernels_base", {0: "scalars", 8: "scalars_b"}),
    "_Z21k_mul_base64_quarters": ("kernels_base", {0: "scalars", 8: "scalars_b"}),
    "_Z10k_mul_coopPKhPKim": ("kernels_coop", {0: "scalars"}),
    "_Z14k_mul_enc_coopPKhS0_m": ("kernels_coop", {0: "scalars"}),
    "_Z15k_mul_base_coopPKhS0_mm": ("kernels_coop", {0: "scalars", 8: "scalars_b"}),
    "_Z11k_sign_coopPKhS0_S0_S0_PKjm": ("kernels_coop", {0: "x (private keys)", 8: "k (nonces)"}),
    "_Z11k_sign_hashPKhS0_S0_PKjm": ("kernels_verify", {0: "x (private keys)", 8: "k (nonces)"}),
    "_Z12k_eddsa_prepPKhS0_PKjm": ("kernels_verify", {0: "seeds"}),
    "_Z19k_pripoly_eval_part": ("kernels_verify", {0: "coefficients of secret polynomials"}),
    "_Z18k_pripoly_eval_sum": ("kernels_verify", {0: "partial values of secret polynomials"}),
    "_Z8k_finishPK": ("kernels_misc", {0: "projective results (a DH shared secret before its encoding)"}),
    "_Z9k_finish4PK": ("kernels_misc", {0: "projective results (a DH shared secret before its encoding)"}),
    "_Z13k_finish_wavePK": ("kernels_coop", {0: "projective results (a DH shared secret before its encoding)", 16: "the same as extended limbs"}),
}
# the windowed-table kernels (mul.algo = 0, radix-16 / -32 fixed base) are selectable cross-checks, not default paths; the public-input
# kernels (verification, decoding, polynomial evaluation at public indices) have nothing to hide

REG_TOKEN = re.compile(r"^(?:[vsa]\[\d+:\d+\]|[vsa]\d+|vcc_lo|vcc_hi|vcc|exec_lo|exec_hi|exec|m0|scc|ttmp\d+)$")


def units(tok):
    """register operand -> list of 32-bit unit names; [] for anything that is not a register"""
    tok = tok.strip()
    neg = re.match(r"^[-|]?(?:abs\(|neg\()?(.*?)\)?[|]?$", tok)
    if neg:
        tok = neg.group(1)
    if not REG_TOKEN.match(tok):
        return []
    if tok == "vcc":
        return ["vcc_lo", "vcc_hi"]
    if tok == "exec":
        return ["exec_lo", "exec_hi"]
    m = re.match(r"^([vsa])\[(\d+):(\d+)\]$", tok)
    if m:
        return [f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    return [tok]


def split_operands(rest):
    """operand list of an instruction line (modifiers like offset:16, sc0, row_shr:1 are dropped)"""
    ops = []
    for part in rest.split(","):
        part = part.strip()
        if not part:
            continue
        first = part.split()[0]
        ops.append(first)
        for extra in part.split()[1:]:          # 'v1 offset:16' / 'off sc0 sc1'
            if REG_TOKEN.match(extra):
                ops.append(extra)
    return ops


class Insn:
    __slots__ = ("idx", "text", "op", "ops", "mods", "dst", "src", "addr", "kind", "partial", "lane_sel", "flows")

    def __init__(self, idx, text):
        self.idx, self.text = idx, text
        sp = text.split(None, 1)
        self.op = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", sp[0])
        rest = sp[1] if len(sp) > 1 else ""
        self.mods = rest
        self.ops = split_operands(rest)
        self.dst, self.src, self.addr, self.lane_sel = [], [], [], []
        self.kind = "alu"
        self.partial = False
        self.flows = None          # per-destination sources where "every destination depends on every source" is too coarse
        self.classify(sp[0])

    def R(self, i):
        return units(self.ops[i]) if i < len(self.ops) else []

    def classify(self, raw):
        op, ops = self.op, self.ops
        allregs = lambda lo=0: [u for i in range(lo, len(ops)) for u in units(ops[i])]
        if op in ("buffer_wbl2", "buffer_inv", "buffer_wbinvl1", "buffer_gl0_inv", "buffer_gl1_inv", "s_waitcnt", "s_nop", "s_barrier", "s_setprio", "s_sleep", "s_endpgm", "s_sethalt", "s_icache_inv", "s_dcache_wb", "s_code_end", "s_clause", "s_setreg_b32", "s_setreg_imm32_b32") \
                or op.startswith("s_waitcnt"):
            self.kind = "none" if op != "s_endpgm" else "end"
        elif op == "s_branch":
            self.kind = "jump"
        elif op.startswith("s_cbranch_"):
            self.kind = "cbranch"
            self.src = {"scc": ["scc"], "vcc": ["vcc_lo", "vcc_hi"], "exe": ["exec_lo", "exec_hi"]}[op[len("s_cbranch_"):][:3]]
        elif op in ("s_setpc_b64", "s_swappc_b64", "s_call_b64", "s_rfe_b64"):
            self.kind = "indirect"
        elif op.startswith("s_load_") or op.startswith("s_buffer_load_"):
            self.kind = "sload"
            self.dst, self.addr = self.R(0), self.R(1) + self.R(2)
        elif op in ("s_memtime", "s_memrealtime", "s_getpc_b64"):
            self.dst = self.R(0)
        elif op.startswith("s_cmp") or op.startswith("s_bitcmp"):
            self.dst, self.src = ["scc"], allregs()
        elif op.startswith("s_"):
            self.dst = self.R(0)
            self.src = allregs(1)
            base = op
            if "saveexec" in base:
                self.dst = self.dst + ["exec_lo", "exec_hi"]
                self.src = self.src + ["exec_lo", "exec_hi"]
            if not (base.startswith("s_mov") or base.startswith("s_movk") or base.startswith("s_cmov") or base.startswith("s_cselect") or base.startswith("s_getreg")
                    or base.startswith("s_mul") or base.startswith("s_ff1") or base.startswith("s_flbit") or base.startswith("s_brev") or base.startswith("s_bcnt0") is False and False):
                self.dst = self.dst + ["scc"]
            if base.startswith(("s_cselect", "s_cmov", "s_addc", "s_subb")):
                self.src = self.src + ["scc"]
        elif op.startswith(("global_load", "flat_load", "scratch_load", "buffer_load")):
            self.kind = "vload_scratch" if op.startswith("scratch_") else "vload"
            self.dst, self.addr = self.R(0), allregs(1)
        elif op.startswith(("global_store", "flat_store", "scratch_store", "buffer_store")):
            self.kind = "vstore_scratch" if op.startswith("scratch_") else "vstore"
            if op.startswith("scratch_"):          # scratch_store_dword off|vaddr, vdata, off|saddr
                self.addr, self.src = self.R(0) + self.R(2), self.R(1)
            else:                                   # global_store_dword vaddr, vdata, off|saddr
                self.addr, self.src = self.R(0) + self.R(2), self.R(1)
        elif op.startswith(("global_atomic", "flat_atomic", "buffer_atomic")):
            self.kind = "vatomic"
            returning = bool(re.search(r"\bsc0\b|\bglc\b", self.mods))
            if returning:
                self.dst, self.addr, self.src = self.R(0), self.R(1) + self.R(3), self.R(2)
            else:
                self.addr, self.src = self.R(0) + self.R(2), self.R(1)
        elif op.startswith(("ds_bpermute", "ds_permute", "ds_swizzle")):
            self.kind = "lanemove"
            self.dst = self.R(0)
            if op.startswith("ds_swizzle"):
                self.src = self.R(1)
            else:
                self.lane_sel, self.src = self.R(1), self.R(2)
        elif op.startswith(("ds_read", "ds_load")):
            self.kind = "ldsload"
            self.dst, self.addr = self.R(0), self.R(1)
        elif op.startswith(("ds_write", "ds_store")):
            self.kind = "ldsstore"
            self.addr, self.src = self.R(0), allregs(1)
        elif op.startswith("ds_"):                  # LDS atomics and the like
            self.kind = "ldsrmw"
            self.dst, self.addr, self.src = self.R(0), self.R(1), allregs(2)
        elif op.startswith(("v_permlane16_swap", "v_permlane32_swap")):
            # gfx950: rows (or halves) of the two registers are exchanged in place — a fixed pattern, no lane-select operand: both registers are
            # written and each takes data from both (the row moves of the one-item-per-wavefront kernels since round 6)
            self.kind = "rowswap"
            self.dst = self.R(0) + self.R(1)
            self.src = self.R(0) + self.R(1)
        elif op.startswith("v_cmpx"):
            self.dst, self.src = ["exec_lo", "exec_hi"] + self.R(0), allregs(1) + ["exec_lo", "exec_hi"]
        elif op.startswith("v_cmp"):
            self.dst, self.src = self.R(0), allregs(1)
        elif op.startswith("v_readlane") and len(ops) > 2 and re.fullmatch(r"\d+", ops[2]) and self.R(1):
            # lane of a VGPR named by an immediate: when the VGPR is one of the compiler's SGPR spill registers (written by v_writelane
            # with an immediate lane, see parse()) its lanes are tracked one by one; a lane of a data register carries the register's taint
            self.dst, self.src = self.R(0), [f"{self.R(1)[0]}.{ops[2]}", self.R(1)[0]]
        elif op.startswith(("v_readfirstlane", "v_readlane")):
            self.dst, self.src = self.R(0), allregs(1)
        elif op.startswith("v_writelane") and len(ops) > 2 and re.fullmatch(r"\d+", ops[2]) and self.R(0):
            self.dst, self.src = [f"{self.R(0)[0]}.{ops[2]}"], self.R(1)
        elif op.startswith("v_writelane"):
            self.dst, self.src, self.partial = self.R(0), allregs(1), True
        elif op.startswith(("v_movrel", "v_movreld", "v_movrels")):
            self.kind = "movrel"
            self.dst, self.src = self.R(0), allregs(1) + ["m0"]
        elif op.startswith("v_"):
            two = ("_co_" in raw) or raw.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale"))
            self.dst = self.R(0) + (self.R(1) if two else [])
            self.src = allregs(2 if two else 1)
            if raw.endswith(("_dpp", "_sdwa")) or "row_" in self.mods or "quad_perm" in self.mods or "dst_sel" in self.mods:
                self.partial = True             # a DPP / SDWA result may keep bits or lanes of the old value
            if op.startswith(("v_mac_", "v_fmac_", "v_dot")) and not two:
                self.src = self.src + self.R(0)  # accumulate into the destination
            # 64-bit results whose LOW word cannot depend on the high word of an addend: the compiler computes 32-bit indices with
            # v_mad_u64_u32 on a register pair whose upper half is whatever was there
            if raw.startswith(("v_mad_u64_u32", "v_mad_i64_i32")) and len(self.R(0)) == 2:
                lo, hi = self.R(0)
                a, b, c = self.R(2), self.R(3), self.R(4)
                self.flows = {lo: a + b + c[:1], hi: a + b + c}
                for u in self.R(1):
                    self.flows[u] = a + b + c
            elif raw.startswith("v_lshl_add_u64") and len(self.R(0)) == 2:
                lo, hi = self.R(0)
                a, sh, b = self.R(1), self.R(2), self.R(3)
                self.flows = {lo: a[:1] + sh + b[:1], hi: a + sh + b}
        else:
            self.kind = "unknown"


def unit_asm(unit):
    """gfx950 assembly of a kernel unit of the library (by name) or of any .hip file (by path)"""
    with tempfile.NamedTemporaryFile(suffix=".s", delete=False) as f:
        out = f.name
    src = unit if unit.endswith(".hip") else os.path.join(CSRC, unit + ".hip")
    # -DKYB_CROSSCHECK: the product's kernels AND the variants of the cross-check build (same sources: the product's kernels are compiled identically)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DKYB_CROSSCHECK", "-S", "--cuda-device-only", "-o", out, src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{unit}: {r.stderr[-2000:]}")
    txt = open(out).read()
    os.unlink(out)
    return txt


def kernel_body(txt, prefix):
    """-> (mangled name, instruction text, first SGPR of the kernarg segment pointer)"""
    m = re.search(r"^(%s\w*):.*?\n(.*?)\n\s*\.amdhsa_kernel \1\n(.*?)\.end_amdhsa_kernel" % re.escape(prefix), txt, re.S | re.M)
    if not m:
        return None, None, None
    desc = m.group(3)
    on = lambda key: re.search(r"\.amdhsa_user_sgpr_%s 1" % key, desc) is not None
    # user SGPRs are handed out in this order (AMDGPU code-object v3+): what comes before the kernarg pointer shifts it
    first = (4 if on("private_segment_buffer") else 0) + (2 if on("dispatch_ptr") else 0) + (2 if on("queue_ptr") else 0)
    if not on("kernarg_segment_ptr"):
        first = None
    return m.group(1), m.group(2), first


def parse(body):
    """-> (instructions, label -> index of the next instruction)"""
    insns, labels = [], {}
    for ln in body.split("\n"):
        code = ln.split(";")[0].strip() if not ln.strip().startswith(";;#") else ""
        if not code or code.startswith(("//", ".")) and not re.match(r"^\.LBB\d+_\d+:", code):
            if re.match(r"^\.LBB\d+_\d+:", code or ""):
                pass
            else:
                continue
        lab = re.match(r"^(\.LBB\d+_\d+):", code)
        if lab:
            labels[lab.group(1)] = len(insns)
            continue
        if code.endswith(":"):
            continue
        insns.append(Insn(len(insns), code))
    spill_regs = {i_.dst[0].split(".")[0] for i_ in insns if i_.op.startswith("v_writelane") and i_.dst and "." in i_.dst[0]}
    for i_ in insns:
        if i_.op.startswith("v_readlane") and len(i_.src) == 2 and "." in i_.src[0]:
            i_.src = [i_.src[0]] if i_.src[1] in spill_regs else [i_.src[1]]
    return insns, labels


def analyse(name, body, secret_offsets, karg_sgpr):
    insns, labels = parse(body)
    n = len(insns)
    # successors
    succ = [[] for _ in range(n)]
    for i, ins in enumerate(insns):
        tgt = None
        if ins.kind in ("jump", "cbranch"):
            t = ins.ops[0] if ins.ops else None
            tgt = labels.get(t)
        if ins.kind == "jump":
            if tgt is not None:
                succ[i].append(tgt)
        elif ins.kind == "cbranch":
            if tgt is not None:
                succ[i].append(tgt)
            if i + 1 < n:
                succ[i].append(i + 1)
        elif ins.kind in ("end", "indirect"):
            pass
        elif i + 1 < n:
            succ[i].append(i + 1)
    # kernarg base: the SGPR pair the kernel descriptor says (valid until the kernel overwrites it: checked per load below)
    # (whether the pair still holds it at a given instruction is part of the dataflow state: a long-branch expansion on ONE path reuses it)
    karg = (f"s{karg_sgpr}", f"s{karg_sgpr + 1}") if karg_sgpr is not None else None
    # long-branch expansions: s_getpc_b64 / s_add_u32 (LABEL - .Lpost_getpc) / s_addc_u32 / s_setpc_b64 is a direct jump to LABEL
    for i, ins in enumerate(insns):
        if ins.kind == "indirect" and ins.op == "s_setpc_b64":
            for back in insns[max(0, i - 4):i]:
                m_ = re.search(r"\((\.LBB\d+_\d+)-\.Lpost_getpc\d+\)&", back.text)
                if m_ and m_.group(1) in labels:
                    ins.kind = "jump"
                    ins.ops = [m_.group(1)]
                    break
    # dataflow: state = (tainted units, pointer-to-secret units); per-instruction IN states, worklist to a fixed point
    IN = [None] * n
    IN[0] = (frozenset(), frozenset(), frozenset(karg or ()), frozenset(), frozenset())
    work = [0]
    MEM_LDS, MEM_SCR = "<lds>", "<scratch>"

    def transfer(ins, st):
        T, P, K, E, A = set(st[0]), set(st[1]), set(st[2]), set(st[3]), set(st[4])
        ret = lambda: (T, P, K, E, A)
        # Wave-aggregated atomics (the compiler's atomic optimizer): ONE lane — the lowest active one — performs the returning atomic inside
        # an exec-masked block, and v_readfirstlane right behind the block broadcasts its result.  On the path around the block the register
        # holds stale data, but v_readfirstlane reads exactly the lane that did the atomic: the value is the atomic's (public) result.
        if ins.op.startswith("v_readfirstlane") and ins.src and all(u in A for u in ins.src):
            for u in ins.dst:
                T.discard(u); P.discard(u); K.discard(u); E.discard(u)
            return ret()
        if ins.kind == "vatomic" and ins.dst:
            for u in ins.dst:
                T.discard(u); P.discard(u); A.add(u)
            return ret()
        for u in ins.dst:
            A.discard(u)
        # EXEC save / restore idioms of structured control flow.  `s_and_saveexec sD, sC` leaves the OLD exec in sD (it does not depend on the
        # condition sC) and narrows exec; the matching `s_or_b64 exec, exec, sD` puts the old value back: exec is then exactly what sD holds.
        if "saveexec" in ins.op:
            sd = units(ins.ops[0]) if ins.ops else []
            cond_t = any(u in T for u in (units(ins.ops[1]) if len(ins.ops) > 1 else []))
            exec_t = "exec_lo" in T or "exec_hi" in T
            for u in sd:
                (T.add if exec_t else T.discard)(u); P.discard(u); K.discard(u); E.add(u)
            for u in ("exec_lo", "exec_hi"):
                (T.add if (exec_t or cond_t) else T.discard)(u)
            T.add("scc") if (exec_t or cond_t) else T.discard("scc")
            return ret()
        if ins.op in ("s_or_b64", "s_mov_b64") and ins.ops and ins.ops[0] == "exec":
            srcs = [o for o in ins.ops[1:] if o != "exec"]
            if len(srcs) == 1 and units(srcs[0]) and all(u in E for u in units(srcs[0])):
                saved_t = any(u in T for u in units(srcs[0]))
                for u in ("exec_lo", "exec_hi"):
                    (T.add if saved_t else T.discard)(u)
                if ins.op == "s_or_b64":
                    T.add("scc") if saved_t else T.discard("scc")
                return ret()
        for u in ins.dst:
            E.discard(u)
        base_is_karg = ins.kind == "sload" and karg is not None and all(u in K for u in karg) and tuple(units(ins.ops[1])) == karg
        for u in ins.dst:
            K.discard(u)
        src_t = any(u in T for u in ins.src)
        lane_t = any(u in T for u in ins.lane_sel)
        addr_p = any(u in P for u in ins.addr)
        k = ins.kind
        if k == "sload":
            # kernarg load: mark the destination units that receive a secret pointer
            dst = ins.dst
            off = None
            if base_is_karg and len(ins.ops) > 2:
                try:
                    off = int(ins.ops[2], 0)
                except ValueError:
                    off = None
            for u in dst:
                T.discard(u); P.discard(u)
            if off is not None:
                for j, u in enumerate(dst):
                    byte = off + 4 * j
                    if (byte & ~7) in secret_offsets:
                        P.add(u)
            elif addr_p:          # a scalar load THROUGH a secret pointer: secret data in SGPRs
                T.update(dst)
            return ret()
        if k in ("vload", "vload_scratch", "ldsload", "ldsrmw"):
            taint = addr_p or (k == "vload_scratch" and MEM_SCR in T) or (k in ("ldsload", "ldsrmw") and MEM_LDS in T)
            for u in ins.dst:
                P.discard(u)
                (T.add if taint else T.discard)(u)
            if k == "ldsrmw" and src_t:
                T.add(MEM_LDS)
            return ret()
        if k in ("vstore", "vatomic"):
            if k == "vatomic":
                for u in ins.dst:
                    T.discard(u); P.discard(u)
            return ret()
        if k == "vstore_scratch":
            if src_t:
                T.add(MEM_SCR)
            return ret()
        if k == "ldsstore":
            if src_t:
                T.add(MEM_LDS)
            return ret()
        if k == "lanemove":
            for u in ins.dst:
                P.discard(u)
                (T.add if (src_t or lane_t) else T.discard)(u)
            return ret()
        if k in ("alu", "movrel"):
            src_p = any(u in P for u in ins.src)
            if ins.flows is not None and not ins.partial:
                t_new = {u: any(x in T for x in srcs) for u, srcs in ins.flows.items()}
                p_new = {u: any(x in P for x in srcs) for u, srcs in ins.flows.items()}
                for u in ins.dst:
                    (T.add if t_new.get(u, src_t) else T.discard)(u)
                    (P.add if p_new.get(u, src_p) else P.discard)(u)
                return ret()
            for u in ins.dst:
                if ins.partial:
                    if src_t:
                        T.add(u)
                    if src_p:
                        P.add(u)
                else:
                    (T.add if src_t else T.discard)(u)
                    (P.add if src_p else P.discard)(u)
            return ret()
        return ret()

    while work:
        i = work.pop()
        out = transfer(insns[i], IN[i])
        out = tuple(frozenset(x) for x in out)
        for j in succ[i]:
            if IN[j] is None:
                IN[j] = out
                work.append(j)
            else:
                merged = (IN[j][0] | out[0], IN[j][1] | out[1], IN[j][2] & out[2], IN[j][3] & out[3], IN[j][4] | out[4])
                if merged != IN[j]:
                    IN[j] = merged
                    work.append(j)
    # sinks
    viol = []
    counts = {"instructions": n, "branches": 0, "memory_accesses": 0, "lane_moves": 0, "row_swaps": 0, "secret_loads": 0, "unreached": sum(1 for s_ in IN if s_ is None)}
    for ins in insns:
        st = IN[ins.idx]
        if st is None:
            continue
        T, P = st[0], st[1]
        exec_t = "exec_lo" in T or "exec_hi" in T
        if ins.kind == "unknown":
            viol.append((ins.idx, "instruction the checker does not model", ins.text))
        if ins.kind == "indirect":
            viol.append((ins.idx, "indirect jump / call", ins.text))
        if ins.kind == "cbranch":
            counts["branches"] += 1
            if any(u in T for u in ins.src):
                viol.append((ins.idx, "branch on secret-dependent condition", ins.text))
        if ins.kind in ("vload", "vstore", "vatomic", "vload_scratch", "vstore_scratch", "ldsload", "ldsstore", "ldsrmw", "sload"):
            counts["memory_accesses"] += 1
            if any(u in T for u in ins.addr):
                viol.append((ins.idx, "memory address depends on a secret", ins.text))
            if exec_t and ins.kind != "sload":
                viol.append((ins.idx, "memory access under a secret-dependent EXEC mask", ins.text))
            if ins.kind in ("vload", "sload") and any(u in P for u in ins.addr):
                counts["secret_loads"] += 1
        if ins.kind == "lanemove":
            counts["lane_moves"] += 1
        if ins.kind == "rowswap":
            counts["row_swaps"] += 1
        if ins.kind == "movrel" and "m0" in T:
            viol.append((ins.idx, "register index (M0) depends on a secret", ins.text))
    return {"kernel": name, "counts": counts, "violations": viol, "kernarg_base": karg}


def check_all(select=None, verbose=False, table=None):
    """table: {kernel-name prefix: (unit name or .hip path, {kernarg offset: description})}; default = the library's secret-handling kernels"""
    wanted = {k: v for k, v in (table or SECRET_ARGS).items() if not select or select in k}
    need_units = sorted({u for u, _ in wanted.values()})
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        texts = dict(zip(need_units, ex.map(unit_asm, need_units)))
    results = []
    for prefix, (unit, offs) in wanted.items():
        name, body, karg_sgpr = kernel_body(texts[unit], prefix)
        if body is None:
            results.append({"kernel": prefix, "error": "kernel not found in " + unit, "violations": [(0, "kernel not found", prefix)], "counts": {}})
            continue
        r = analyse(name, body, offs, karg_sgpr)
        r["unit"], r["secret_args"] = unit, offs
        if r["counts"]["secret_loads"] == 0:
            r["violations"].append((0, "no load through a secret argument was recognised: the source model is broken", prefix))
        results.append(r)
    return results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default=None)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    res = check_all(a.kernel, a.verbose)
    bad = 0
    for r in res:
        c = r.get("counts", {})
        print(f"{r['kernel'][:60]:<60} {c.get('instructions', 0):>6} instr  {c.get('branches', 0):>3} branches  {c.get('memory_accesses', 0):>4} memory ops  "
              f"{c.get('lane_moves', 0):>4} lane moves  {c.get('secret_loads', 0):>2} secret loads  ->  {'ok' if not r['violations'] else str(len(r['violations'])) + ' VIOLATIONS'}")
        for idx, what, text in r["violations"][: (1000 if a.verbose else 8)]:
            print(f"      [{idx}] {what}: {text}")
        bad += len(r["violations"])
    if a.json:
        json.dump([{**r, "violations": [list(v) for v in r["violations"]], "kernarg_base": list(r.get("kernarg_base") or [])} for r in res], open(a.json, "w"), indent=1)
    print(f"{len(res)} kernels checked, {bad} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
