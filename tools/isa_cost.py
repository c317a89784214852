"""VALU issue cycles per basic block of the kernels in a -save-temps ISA file, priced with the issue costs that
tools/probes/valu_rate.hip measures on gfx950 with a full SIMD (profiles/valu_issue_cost.txt): 2 cycles for f32
add / sub / mul / fma / fmac, and / or / xor / not, add / sub u32, mov, lshrrev, ashrrev and bitop3; 8 for
transcendentals; 4 for everything else (min / max, compares, cndmask, conversions, shifts left, bfe / bfi / perm,
three-operand integer forms, DPP and SDWA forms, packed f32 and 16-bit forms, f64 add / mul / fma).
  python tools/isa_cost.py <file.s> <kernel symbol>          per-block table of one kernel
  python tools/isa_cost.py <file.s> --json <out.json>        per-kernel totals of all k_* kernels: instructions by
                                                             class and the mean issue cost of the instructions that
                                                             the counters do not break out (bench.py prices with it)"""
import json
import re
import subprocess
import sys

FAST = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_bitop3_b32", "v_add_co_u32",
        "v_addc_co_u32", "v_accvgpr_read_b32", "v_accvgpr_write_b32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}
# what the SQ_INSTS_VALU_{FMA,MUL,ADD}_F32, _CVT and _TRANS_F32 counters break out
COUNTED = {"fma": {"v_fma_f32", "v_fmac_f32"}, "mul": {"v_mul_f32"}, "add": {"v_add_f32", "v_sub_f32", "v_subrev_f32"}}


def base_of(op):
    return re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)


def cost(op):
    base = base_of(op)
    if op.endswith(("_sdwa", "_dpp")):
        return 4
    if base in FAST:
        return 2
    if base in TRANS:
        return 8
    if base in ("v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_rcp_f64", "v_sqrt_f64"):
        return 8  # (not measured)
    return 4


def klass(op):
    base = base_of(op)
    if not op.endswith(("_sdwa", "_dpp")):
        for k, s in COUNTED.items():
            if base in s:
                return k
    if base in TRANS:
        return "trans"
    if base.startswith("v_cvt_"):
        return "cvt"
    return "rest"


def kernels(lines):
    """name -> list of (block, opcode) of the kernel's instructions"""
    out, name, block = {}, None, None
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, block = m.group(1), "entry"
            out[name] = []
            continue
        if name is None:
            continue
        if l.strip() == "s_endpgm" or l.startswith("\t.end_amdhsa_kernel"):
            name = None
            continue
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            block = m.group(1)
            continue
        m = re.match(r"^\s+([a-z][a-z0-9_]+)", l)
        if m:
            out[name].append((block, m.group(1)))
    return out


def is_valu(op):
    return op.startswith("v_") and not op.startswith(("v_readlane", "v_readfirstlane", "v_writelane"))


def table(ins):
    blocks, order = {}, []
    for block, op in ins:
        b = blocks.setdefault(block, {"valu": 0, "cyc": 0, "salu": 0, "vmem": 0, "lds": 0, "ops": {}})
        if block not in order:
            order.append(block)
        if is_valu(op):
            b["valu"] += 1
            b["cyc"] += cost(op)
            b["ops"][op] = b["ops"].get(op, 0) + 1
        elif op.startswith("s_"):
            b["salu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            b["vmem"] += 1
        elif op.startswith("ds_"):
            b["lds"] += 1
    tv = tc = 0
    for k in order:
        b = blocks[k]
        tv += b["valu"]
        tc += b["cyc"]
        if b["valu"] >= 8:
            top = sorted(b["ops"].items(), key=lambda kv: -kv[1] * cost(kv[0]))[:6]
            print(f"{k:12s} valu {b['valu']:4d} cycles {b['cyc']:5d} salu {b['salu']:3d} vmem {b['vmem']:2d} lds {b['lds']:2d}  " +
                  " ".join(f"{o}x{n}" for o, n in top))
    print(f"total valu {tv} cycles {tc} ({tc / max(tv, 1):.2f} per instruction)")


def summary(ins):
    by = {}
    for _, op in ins:
        if is_valu(op):
            k = klass(op)
            d = by.setdefault(k, {"n": 0, "cycles": 0})
            d["n"] += 1
            d["cycles"] += cost(op)
    n = sum(d["n"] for d in by.values())
    c = sum(d["cycles"] for d in by.values())
    rest = by.get("rest", {"n": 0, "cycles": 0})
    return {"valu": n, "cycles": c, "mean_cost": round(c / max(n, 1), 3), "classes": by,
            "rest_mean_cost": round(rest["cycles"] / max(rest["n"], 1), 3)}


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout.split("\n")
        return {n: re.sub(r"^void ", "", d).split("(")[0] for n, d in zip(names, out)}
    except Exception:
        return {n: n for n in names}


def main():
    path = sys.argv[1]
    ks = kernels(open(path).read().splitlines())
    if sys.argv[2] == "--json":
        names = [n for n in ks if re.match(r"_Z\d+k_", n)]
        dm = demangle(names)
        out = {"source": "static instruction mix of the kernels' ISA (every instruction counted once), priced per class; "
                         "rest = instructions that no SQ_INSTS_VALU_* class counter breaks out",
               "issue_cost_cycles": {"fma": 2, "mul": 2, "add": 2, "cvt": 4, "trans": 8},
               "kernels": {dm[n]: summary(ks[n]) for n in names}}
        json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
        for n in names:
            s = out["kernels"][dm[n]]
            print(f"{dm[n]:40s} valu {s['valu']:5d} mean cost {s['mean_cost']:.2f} rest {s['rest_mean_cost']:.2f}")
    else:
        table(ks[sys.argv[2]])


main()
