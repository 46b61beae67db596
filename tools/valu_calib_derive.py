"""gpurun_out/calib/{plain,pmc}.txt (tools/valu_calib.sh) -> profiles/r04_valu_calibration.{txt,json}: what the two
"VALU utilisation" formulas read on streams whose instruction count is known, at the launch shape of k_icp_pipe
(one 1024-thread workgroup per CU = 4 waves per SIMD) and at 1 and 2 waves per SIMD.
  A = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x waves per SIMD       (round 3's "issue_frac")
  B = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x cycles of the launch)   (VERDICT r3's first-principles figure)
usage: python tools/valu_calib_derive.py [gpurun_out/calib] [profiles]"""
import ast, json, os, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/calib"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"
plain = [json.loads(l) for l in open(os.path.join(src, "plain.txt")) if l.startswith("{")]
dev, runs = plain[0], plain[1:]
pmc = {}
for l in open(os.path.join(src, "pmc.txt")):
    if l.startswith("==") or " wg " not in l:
        continue
    name, rest = l.split(" wg ", 1)
    wg, d = rest.split(" ", 1)
    pmc.setdefault((name, int(wg)), {}).update({k: float(v) for k, v in ast.literal_eval(d.strip()).items()})
streams = {}
for r in runs:
    wps = r["waves_per_simd"]
    c = pmc.get((r["stream"], wps * 256), {})
    waves = 1024 * wps
    e = {"mean_wave_cycles": r["mean_wave_cycles"], "event_ms": r["event_ms"],
         "clock_GHz_under_load": r["mean_wave_cycles"] / (r["event_ms"] * 1e6)}
    if c.get("SQ_WAVE_CYCLES"):
        cyc = c["SQ_WAVE_CYCLES"] * 4.0 / waves          # SQ_WAVE_CYCLES counts quad-cycles, summed over the waves
        e.update({"SQ_INSTS_VALU": c.get("SQ_INSTS_VALU"), "SQ_ACTIVE_INST_VALU": c.get("SQ_ACTIVE_INST_VALU"), "SQ_WAVE_CYCLES": c["SQ_WAVE_CYCLES"],
                  "wave_cycles_from_SQ": cyc,
                  "A_active_over_wave_cycles_x_waves": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"] * wps if c.get("SQ_ACTIVE_INST_VALU") else None,
                  "B_insts_x2_over_simd_cycles": c["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cyc) if c.get("SQ_INSTS_VALU") else None,
                  "valu_per_cycle_per_simd": c["SQ_INSTS_VALU"] / (1024.0 * cyc) if c.get("SQ_INSTS_VALU") else None,
                  "valu_per_ns_per_simd": c["SQ_INSTS_VALU"] / 1024.0 / (r["event_ms"] * 1e6) if c.get("SQ_INSTS_VALU") else None,
                  "wait_any_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAIT_ANY") else None,
                  "wait_inst_any_frac": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAIT_INST_ANY") else None})
    streams.setdefault(r["stream"], {})["waves_per_simd_%d" % wps] = e
sat = streams["fma_indep"]["waves_per_simd_4"]
mix = streams["mix_search"]["waves_per_simd_4"]
out = {"device": dev, "what": __doc__.split("\n")[0], "streams": streams,
       "saturation": {"stream": "fma_indep, 4 waves per SIMD (k_icp_pipe's launch shape)",
                      "A": sat["A_active_over_wave_cycles_x_waves"], "B": sat["B_insts_x2_over_simd_cycles"],
                      "valu_per_cycle_per_simd": sat["valu_per_cycle_per_simd"], "valu_per_ns_per_simd": sat["valu_per_ns_per_simd"],
                      "clock_GHz_under_load": sat["clock_GHz_under_load"]},
       "saturation_search_mix": {"stream": "mix_search (the grid walk's point loop: 2 ds_read_b128 + 2 distances + 2 64-bit key minima per trip), 4 waves per SIMD",
                                 "A": mix["A_active_over_wave_cycles_x_waves"], "B": mix["B_insts_x2_over_simd_cycles"],
                                 "valu_per_cycle_per_simd": mix["valu_per_cycle_per_simd"], "valu_per_ns_per_simd": mix["valu_per_ns_per_simd"]},
       "reading": "Neither formula reads 1.0 at saturation: A reads %.2f (SQ_ACTIVE_INST_VALU counts one quad-cycle per vector instruction, "
                  "whatever the pipe needs) and B reads %.2f (a SIMD retires a wave64 f32 instruction every %.2f cycles with four waves to "
                  "pick from, not every 2).  Divide a kernel's A or B by these to get the fraction of the saturated rate." %
                  (sat["A_active_over_wave_cycles_x_waves"], sat["B_insts_x2_over_simd_cycles"], 1.0 / sat["valu_per_cycle_per_simd"])}
json.dump(out, open(os.path.join(dst, "r04_valu_calibration.json"), "w"), indent=1)
with open(os.path.join(dst, "r04_valu_calibration.txt"), "w") as f:
    f.write("# tools/valu_calib.sh on MI355X: tools/bin/valu_calib plain (s_memtime cycles, HIP events), then under four rocprofv3 --pmc passes\n")
    f.write("# derived figures: profiles/r04_valu_calibration.json (tools/valu_calib_derive.py)\n## plain\n")
    f.write(open(os.path.join(src, "plain.txt")).read())
    f.write("## pmc\n")
    f.write(open(os.path.join(src, "pmc.txt")).read())
print(json.dumps(out["saturation"], indent=1)); print(json.dumps(out["saturation_search_mix"], indent=1)); print(out["reading"])
