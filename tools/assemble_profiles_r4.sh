#!/bin/bash
# after tools/final_measure_r4.sh a + b and tools/r4_collect_modes.sh: copy the merged gpurun_out/ results into profiles/
# (tracked) and rebuild profiles/traffic.json with every mode stamped with the current source hash
S=gpurun_out/r04_final; P=profiles
cp gpurun_out/traffic_all_modes.json $P/traffic.json
python tools/make_traffic_json.py $S/pmc_traffic.json $S/pmc_sq.json c2 $P/traffic.json 4096 > /dev/null
cp $S/bench.json $P/r04_final_bench.json; cp $S/kernel_stats.csv $P/r04_final_kernel_stats.csv; cp $S/pmc_sq.json $P/r04_final_pmc_sq.json; cp $S/pmc_traffic.json $P/r04_final_pmc_traffic.json
for m in match rotate3 batch256 forcecomm align; do cp $S/bench_$m.json $P/r04_final_${m}_bench.json; done
for m in ref c4 c5 c3; do cp ${S}_$m/bench.json $P/r04_final_${m}_bench.json; cp ${S}_$m/kernel_stats.csv $P/r04_final_${m}_kernel_stats.csv; done
cp ${S}_c3/pmc_sq.json $P/r04_final_c3_pmc_sq.json; cp ${S}_c3/pmc_traffic.json $P/r04_final_c3_pmc_traffic.json
cp $S/align_kernel_stats.csv $P/r04_final_align_kernel_stats.csv; cp $S/align_pmc_sq.json $P/r04_final_align_pmc_sq.json; cp $S/align_pmc_traffic.json $P/r04_final_align_pmc_traffic.json
cp $S/phase_counters.txt $P/r04_final_phase_counters.txt; cp $S/coexec_probe.txt $P/r04_coexec_probe.txt; cp $S/match_overlap_probe.txt $P/r04_match_overlap_probe.txt; cp $S/soak.txt $P/r04_final_soak.txt; cp $S/stage_latency.txt $P/r04_final_stage_latency.txt; cp $S/latency_probe.txt $P/r04_final_latency_probe.txt
python - <<'PY'
import json,sys
sys.path.insert(0,'jetracer-orbslam2_amd'); import orbfe
t=json.load(open('profiles/traffic.json')); h=orbfe.source_hash()
print({k:("ok" if v.get('csrc_sha256')==h else "STALE") for k,v in t.items() if isinstance(v,dict)})
d=json.load(open('profiles/r04_final_bench.json'))
print("c2 value %.4g ms %.4f fps %.0f" % (d["value"], d["ms_per_step"], d["frames_per_s"]), {k:round(v,4) for k,v in d["stage_ms"].items()}, "roof %.4f ach %.0f" % (d["roofline"]["frac"], d["roofline"]["achieved"]), "stale", d["roofline"]["pmc"]["stale"])
print("path %.4f %.0f gp %.0f" % (d["path_hbm"]["frac_of_8TBps"], d["path_hbm"]["achieved_GBps"], d["matcher_gpairs_per_s"]), "survey %.4g %.4f %.0f fps" % (d["survey_scene"]["value"], d["survey_scene"]["ms_per_step"], d["survey_scene"]["frames_per_s"]), "sustained %.4f" % d["sustained"]["ms_per_step"])
for k,v in d["fixed_modes"].items():
    if isinstance(v,dict): print(k, "%.4g"%v["value"], round(v["ms_per_step"],4), "%.0f fps" % (v["value"]/2000), "describe %.3f" % v["stage_ms"]["describe"], "%.1f%%" % (100*(v["value"]/d["value"]-1)))
print("cpu %.3g (%.3g)" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["single_thread"]["value"]))
for k,e in d["roofline"]["stages"].items(): print(k, e.get("limiter"), {a:round(b,3) for a,b in (e.get("limiter_utilisation") or {}).items()})
a=json.load(open('profiles/r04_final_align_bench.json'))
print("align %.0f fps %.4f ms roof %.4f ach %.0f traffic %.3g px/s %.4g cpu %.0f (%.0f)" % (a["value"], a["ms_per_step"], a["roofline"]["frac"], a["roofline"]["achieved"], a["roofline"]["traffic"] or 0, a["pixels_per_s"], a["cpu_baseline"]["value"], a["cpu_baseline"]["single_thread"]["value"]), a["roofline"]["pmc"])
for m in ("ref","c3","c4","c5"):
    x=json.load(open('profiles/r04_final_%s_bench.json'%m)); print(m, "%.4g"%x["value"], round(x["ms_per_step"],4), "%.4g fps"%x["frames_per_s"], "cpu %.3g"%x["cpu_baseline"]["value"], "path %.3f %.0f" % (x["path_hbm"]["frac_of_8TBps"], x["path_hbm"]["achieved_GBps"]), x.get("matcher_gpairs_per_s"), "stale", x["roofline"]["pmc"]["stale"])
x=json.load(open('profiles/r04_final_match_bench.json'))
for s in x["sizes"]: print(s["n"], round(s["brute_force_256bit"]["ms_per_call"],4), round(s["brute_force_256bit"].get("gpairs_per_s"),0), round(s["reference_32bit_window2"]["ms_per_call"]*1e3,1))
PY
awk -F'",' 'NR>1 && NR<8 {print substr($1,1,40), $2}' profiles/r04_final_kernel_stats.csv; awk -F'",' 'NR>1 && NR<5 {print substr($1,1,40), $2}' profiles/r04_final_align_kernel_stats.csv
grep -v amdgpu profiles/r04_final_stage_latency.txt profiles/r04_final_latency_probe.txt profiles/r04_match_overlap_probe.txt | cut -d: -f2-; tail -3 profiles/r04_coexec_probe.txt
