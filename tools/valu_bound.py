"""The integer-VALU issue bound of the he_mul RNS core, recomputable from committed evidence (dev tool; runs in the container).

    python tools/valu_bound.py profiles/r03/vN_instr_rate.txt profiles/r03/vN_issue_probe.txt [profiles/r03/vM_pmc_summary.json] > profiles/r03/vK_valu_bound.json

1. compiles gpqhe_amd/csrc/engine.hip to gfx950 assembly (hipcc -S, device only) and counts, per kernel of the core, the static
   VALU instruction mix (the kernels are straight-line code: every instruction of the listing runs once per wave);
2. prices every opcode with the issue rate `tools/instr_rate` measured on an MI355X (cycles per wave instruction on one SIMD =
   2 x the "lane-cycles per instr" column: the tool normalises to 128 lanes per clock per CU, a SIMD issues 64 lanes per 4 clocks);
   opcodes the probe does not cover take the rate of their class (see CLASS below), and the table says which ones did;
3. weights the kernels by their dynamic VALU instruction counts (SQ_INSTS_VALU of the PMC pass, when given) into one figure:
   the cycles per VALU wave-instruction this instruction mix needs when nothing but issue is in the way (`static` estimate);
4. reads the run of tools/issue_probe -- the library's own butterfly groups in a register-resident loop on every SIMD -- and
   multiplies by the VALU instructions of the probe's loop bodies (its own assembly): the MEASURED VALU wave-instructions per
   second of this code with no memory traffic, with the shader clock and package power rocm-smi showed meanwhile (the sustained
   runs of tools/gpu_r3_probe.sh).  `issue_bound` is the best sustained run: what bench.py prices the core against.
   (The static estimate is normalised to a nominal 2.4 GHz like tools/instr_rate itself.)
bench.py reads the newest profiles/r*/*valu_bound.json for its `valu_issue` object and flags it stale when the kernel sources
have changed since (sha256 of the two kernel headers)."""
import collections, hashlib, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpqhe_amd", "csrc")
KERNELS = {   # bench.py's kernel names -> mangled-name fragments of the headline instantiations (n = 2^16, wide-split limbs)
    "strided_fwd": "strided_passILi8ELi4ELb0ELb0ENS_3TwWELi8E",
    "strided_inv": "strided_passILi8ELi4ELb1ELb0ENS_3TwWELi8E",
    "tensor_mid": "tensor_mid8INS_3TwWELi8E",
    "keyswitch_mid": "keyswitch_mid8x2INS_3TwWELi8ELb1E",
    "contig_fwd": "contig_passILb0ENS_3TwWEEE",
    "contig_inv": "contig_passILb1ENS_3TwWEEE",
}
# opcode -> the probe line of tools/instr_rate whose rate it takes when it has none of its own
CLASS = [
    (r"v_mad_u64_u32|v_mad_i64_i32", "v_mad_u64_u32 (acc)"),
    (r"v_lshl_add_u64|v_add_co_u32|v_addc_co_u32|v_sub_co_u32|v_subb_co_u32|v_subrev_co_u32|v_subbrev_co_u32", None),   # own lines
    (r"v_cmp_.*_u64|v_cmp_.*_i64", "v_cmp_u64"),
    (r"v_cmp_", "v_cmp_u32"),
    (r"v_cndmask_b32", "v_cndmask_b32 (sgpr mask)"),
    (r"v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64", "v_lshlrev_b64"),
    (r"v_alignbit_b32|v_alignbyte_b32", "v_alignbit_b32"),
    (r"v_bitop3_b32|v_bfi_b32|v_and_or_b32|v_or3_b32|v_xad_u32|v_lshl_or_b32|v_lshl_add_u32|v_add_lshl_u32|v_add3_u32|v_bfe_u32|v_perm_b32|v_min3_u32", "v_add3_u32"),
    (r"v_mul_hi_u32|v_mul_lo_u32|v_mul_u32_u24|v_mad_u32_u24", "v_mul_hi_u32"),
    (r"v_pk_", "v_pk_add_u16"),
    (r"v_accvgpr|v_readlane|v_readfirstlane|v_writelane", "v_mov_b32"),
    (r"v_", "v_and_b32"),      # two-operand 32-bit: and / or / xor / add / sub / mov / shifts / min
]


def rates(path):
    out = {}
    for ln in open(path):
        m = re.match(r"(\S.*?)\s+[0-9.]+ ms\s+([0-9.]+) lane-cycles per statement \((\d+) instr\) -> ([0-9.]+) per instr", ln)
        if m:
            out[m.group(1).strip()] = 2.0 * float(m.group(4))          # cycles per wave instruction on one SIMD
    return out


def price(op, r):
    direct = {"v_add_co_u32": "v_add_co_u32 + v_addc_co_u32", "v_addc_co_u32": "v_add_co_u32 + v_addc_co_u32", "v_sub_co_u32": "v_sub_co_u32 + v_subb_co_u32",
              "v_subb_co_u32": "v_sub_co_u32 + v_subb_co_u32", "v_subrev_co_u32": "v_sub_co_u32 + v_subb_co_u32", "v_subbrev_co_u32": "v_sub_co_u32 + v_subb_co_u32",
              "v_lshl_add_u64": "v_lshl_add_u64", "v_mov_b32": "v_mov_b32", "v_add_u32": "v_add_u32", "v_min_u32": "v_min_u32"}
    base = op.replace("_e32", "").replace("_e64", "").replace("_dpp", "").replace("_sdwa", "")
    if base in direct and direct[base] in r:
        return r[direct[base]], direct[base]
    if base in r:
        return r[base], base
    for pat, line in CLASS:
        if line and re.match(pat, base):
            if line in r:
                return r[line], line
    return r["v_and_b32"], "v_and_b32"


def probe_loops():
    """VALU instructions of the main loop of each probe<MIX> of tools/issue_probe.hip (the loop with the most VALU instructions)."""
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "probe.s")
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", CSRC, "-S", "--cuda-device-only", "-o", asm,
                               os.path.join(ROOT, "tools", "issue_probe.hip")], stderr=subprocess.DEVNULL)
        text = open(asm).read()
    out = {}
    for mix in range(3):
        m = re.search(r"^(_Z5probeILi%dE[^\n:]*):[^\n]*\n(.*?)^\.Lfunc_end" % mix, text, flags=re.M | re.S)
        body, best = m.group(2), []
        for lm in re.finditer(r"^(\.LBB\d+_\d+):[^\n]*\n", body, flags=re.M):
            bm = re.search(r"s_cbranch_\w+ %s\b" % re.escape(lm.group(1)), body[lm.end():])
            if not bm:
                continue
            loop = body[lm.end():lm.end() + bm.start()]
            ops = [l.split()[0] for l in (x.strip() for x in loop.split("\n")) if l and not l.startswith((".", ";")) and not l.endswith(":")]
            valu = [o for o in ops if o.startswith("v_")]
            if len(valu) > len(best):
                best = valu
        out[mix] = best
    return out


def main():
    rate_file, probe_file = sys.argv[1], sys.argv[2]
    pmc_file = sys.argv[3] if len(sys.argv) > 3 else None
    r = rates(rate_file)
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "engine.s")
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", asm, os.path.join(CSRC, "engine.hip")],
                              stderr=subprocess.DEVNULL)
        text = open(asm).read()
    src_hash = hashlib.sha256(b"".join(open(os.path.join(CSRC, f), "rb").read() for f in ("modarith.hpp", "ntt_kernels.hpp"))).hexdigest()[:16]
    out = {"_rates_file": os.path.relpath(rate_file, ROOT), "_kernel_source_sha16": src_hash, "_unit": "cycles per VALU wave-instruction on one SIMD",
           "_head": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip(), "kernels": {}}
    used_class = collections.Counter()
    for name, frag in KERNELS.items():
        m = re.search(r"^(_ZN3gpq\d+%s[^\n:]*):[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(frag), text, flags=re.M | re.S)
        if not m:
            continue
        body = m.group(2)
        ops = [l.split()[0] for l in (x.strip() for x in body.split("\n")) if l and not l.startswith((".", ";")) and not l.endswith(":")]
        valu = collections.Counter(o for o in ops if o.startswith("v_") and not o.startswith("v_mfma"))
        branches = sum(1 for o in ops if o.startswith("s_cbranch"))
        cyc = 0.0
        for o, n in valu.items():
            c, line = price(o, r)
            cyc += c * n
            used_class[(o.replace("_e32", "").replace("_e64", ""), line)] += n
        tot = sum(valu.values())
        out["kernels"][name] = {"valu_insts_per_wave": tot, "issue_cycles_per_wave": round(cyc, 1), "static_cycles_per_valu_inst_at_2400MHz": round(cyc / tot, 3),
                                "multiplies": sum(n for o, n in valu.items() if o.startswith("v_mad_u64") or o.startswith("v_mad_i64")),
                                "branches": branches, "top": dict(valu.most_common(12))}
    out["_pricing"] = {"%s <- %s" % k: n for k, n in used_class.most_common()}
    # the measured probe
    loops = probe_loops()
    probe = {"file": os.path.relpath(probe_file, ROOT), "runs": []}
    for ln in open(probe_file):
        m = re.match(r"mix (\d) (.*?)\s+waves/SIMD (\d+)\s+iters (\d+)\s+cycles/iter/wave mean ([0-9.]+) max ([0-9.]+)\s+kernel ([0-9.]+) ms\s+implied clock ([0-9.]+) MHz", ln)
        if not m:
            continue
        mix, w = int(m.group(1)), int(m.group(3))
        n_valu = len(loops[mix])
        static = sum(price(o, r)[0] for o in loops[mix]) / n_valu
        probe["runs"].append({"mix": m.group(2).strip(), "waves_per_simd": w, "valu_insts_per_iter": n_valu, "multiplies_per_iter": sum(1 for o in loops[mix] if o.startswith("v_mad_u64")),
                              "s_memtime_ticks_per_iter_per_wave_mean_max": [float(m.group(5)), float(m.group(6))],
                              "static_estimate_cycles_per_valu_inst_at_2400MHz": round(static, 3),
                              "valu_wave_insts_per_s": round(256 * 4 * w * n_valu * int(m.group(4)) / (float(m.group(7)) * 1e-3), -7)})
    # the sustained runs (issue_probe long W) with the rocm-smi samples the script put under each of them
    lines = open(probe_file).read().split("\n")
    longs = []
    for i, ln in enumerate(lines):
        m = re.match(r"long mix 2 waves/SIMD (\d+)\s+launches (\d+)\s+iters (\d+)\s+waves (\d+)\s+total ([0-9.]+) ms", ln)
        if not m:
            continue
        w, launches, iters, waves, ms = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5))
        smi = []
        for nxt in lines[i + 1:]:
            sm = re.match(r"\((\d+)Mhz\)\s+([0-9.]+)", nxt)
            if not sm:
                if nxt.startswith("#"):
                    continue
                break
            smi.append((int(sm.group(1)), float(sm.group(2))))
        busy = [x for x in smi if x[1] > 600]            # samples taken while the probe was running
        rate = launches * iters * waves * len(loops[2]) / (ms * 1e-3)
        rec = {"waves_per_simd": w, "valu_wave_insts_per_s": round(rate, -7), "valu_insts_per_iter": len(loops[2]), "total_ms": ms}
        if busy:
            rec["sclk_MHz"] = round(sum(x[0] for x in busy) / len(busy))
            rec["package_W"] = round(sum(x[1] for x in busy) / len(busy))
            rec["cycles_per_valu_inst"] = round(rec["sclk_MHz"] * 1e6 * 1024 / rate, 3)
        longs.append(rec)
    probe["sustained"] = longs
    out["probe"] = probe
    if longs:
        best = max(longs, key=lambda x: x["valu_wave_insts_per_s"])
        out["issue_bound"] = {"valu_wave_insts_per_s": best["valu_wave_insts_per_s"], "cycles_per_valu_inst": best.get("cycles_per_valu_inst"),
                              "sclk_MHz": best.get("sclk_MHz"), "package_W": best.get("package_W"), "waves_per_simd": best["waves_per_simd"],
                              "what": "tools/issue_probe long: the library's own forward group + 8 lazy products + inverse group (wide-split class) in a register-resident "
                                      "loop on every SIMD, no memory traffic: the rate at which this instruction mix issues when nothing else is in the way"}
    if pmc_file:
        pmc = json.load(open(pmc_file))
        wsum = csum = 0.0
        weights = {}
        for kname, v in pmc.items():
            if kname.startswith("_"):
                continue
            key = ("strided_fwd" if "strided_pass<8, 4, false" in kname else "strided_inv" if "strided_pass<8, 4, true" in kname else
                   "tensor_mid" if "tensor_mid8" in kname else "keyswitch_mid" if "keyswitch_mid8x2" in kname else None)
            if key is None or key not in out["kernels"]:
                continue
            launches = 2 if key.startswith("strided") else 1           # a strided pass runs for the tensor stage and for the key switch
            w = v["SQ_INSTS_VALU"] * launches
            weights[key] = w
            wsum += w
            csum += w * out["kernels"][key]["static_cycles_per_valu_inst_at_2400MHz"]
        out["pipeline"] = {"pmc_file": os.path.relpath(pmc_file, ROOT), "pmc_head": pmc.get("_head"), "pmc_chunk": pmc.get("_chunk"), "valu_insts_per_launch_group": weights,
                           "valu_wave_insts_per_he_mul": int(wsum / pmc.get("_chunk", 16)),
                           "static_estimate_cycles_per_valu_inst_at_2400MHz": round(csum / wsum, 3)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
