#!/bin/bash
# after tools/final_measure_r5.sh a, b, c: copy the merged gpurun_out/ results into profiles/ (tracked) and rebuild
# profiles/traffic.json with every mode stamped with the current source hash (each gpurun call starts from the snapshot's stale
# traffic.json, so the entries are re-made here from the PMC summaries the calls brought back)
S=gpurun_out/r05_final; P=profiles
for m in c3 ref c4 c5; do
  python tools/make_traffic_json.py ${S}_$m/pmc_traffic.json ${S}_$m/pmc_sq.json $m $P/traffic.json $(python3 -c "import json;print(json.load(open('${S}_$m/bench.json'))['config']['frames_per_gpu_per_step'])") > /dev/null
done
python tools/make_traffic_json.py $S/pmc_traffic.json $S/pmc_sq.json c2 $P/traffic.json 4096 > /dev/null
python tools/make_align_traffic.py $S/align_pmc_traffic.json 8 1024 $P/traffic.json > /dev/null
cp $S/bench.json $P/r05_final_bench.json; cp $S/kernel_stats.csv $P/r05_final_kernel_stats.csv; cp $S/pmc_sq.json $P/r05_final_pmc_sq.json; cp $S/pmc_traffic.json $P/r05_final_pmc_traffic.json
for m in match match_stream match_tile ingest rotate3 batch256 forcecomm align c2_stream_matcher c2_tile_matcher ref c4 c5; do cp $S/bench_$m.json $P/r05_final_${m}_bench.json; done
for m in ref c4 c5 c3; do cp ${S}_$m/bench.json $P/r05_final_${m}_bench.json; cp ${S}_$m/kernel_stats.csv $P/r05_final_${m}_kernel_stats.csv; done
cp ${S}_c3/pmc_sq.json $P/r05_final_c3_pmc_sq.json; cp ${S}_c3/pmc_traffic.json $P/r05_final_c3_pmc_traffic.json
cp $S/align_kernel_stats.csv $P/r05_final_align_kernel_stats.csv; cp $S/align_pmc_sq.json $P/r05_final_align_pmc_sq.json; cp $S/align_pmc_traffic.json $P/r05_final_align_pmc_traffic.json
cp $S/phase_counters.txt $P/r05_final_phase_counters.txt; cp $S/coexec_probe.txt $P/r05_coexec_probe.txt; cp $S/mfma_fold_probe.txt $P/r05_mfma_fold_probe.txt
cp $S/ingest_probe.txt $P/r05_ingest_probe.txt; cp $S/soak.txt $P/r05_final_soak.txt; cp $S/stage_latency.txt $P/r05_final_stage_latency.txt; cp $S/latency_probe.txt $P/r05_final_latency_probe.txt
{ echo "# SQ counters of the 256-bit brute-force matcher inside the default bench step (tools/r5_match_pmc.sh; rocprofv3 --pmc, separate passes)"; echo "## product: match_tile_kernel (large calls)"; grep match $S/match_pmc_tile.txt; echo "## round 4's form, forced with ORBFE_MATCH=stream: match_expand_kernel (0.289 ms, not shown) + match_mfma_kernel"; grep match $S/match_pmc_stream.txt; } > $P/r05_final_match_pmc.txt
cp gpurun_out/r05_final_matchpmc/kernel_stats_base.csv $P/r05_final_match_tile_kernel_stats.csv
python - <<'PY'
import json,sys
sys.path.insert(0,'jetracer-orbslam2_amd'); import orbfe
t=json.load(open('profiles/traffic.json')); h=orbfe.source_hash()
print({k:("ok" if v.get('csrc_sha256')==h else "STALE") for k,v in t.items() if isinstance(v,dict)})
d=json.load(open('profiles/r05_final_bench.json'))
print("c2 value %.4g ms %.4f fps %.0f" % (d["value"], d["ms_per_step"], d["frames_per_s"]), {k:round(v,4) for k,v in d["stage_ms"].items()}, "roof %.4f ach %.0f" % (d["roofline"]["frac"], d["roofline"]["achieved"]), "stale", d["roofline"]["pmc"]["stale"])
print("path %.4f %.0f gp %.0f" % (d["path_hbm"]["frac_of_8TBps"], d["path_hbm"]["achieved_GBps"], d["matcher_gpairs_per_s"]), "survey %.4g %.4f %.0f fps" % (d["survey_scene"]["value"], d["survey_scene"]["ms_per_step"], d["survey_scene"]["frames_per_s"]), "sustained %.4f over %.2f s" % (d["sustained"]["ms_per_step"], d["sustained"]["seconds"]))
for k,v in d["fixed_modes"].items():
    if isinstance(v,dict): print(k, "%.4g"%v["value"], round(v["ms_per_step"],4), "%.0f fps" % (v["value"]/2000), "describe %.3f" % v["stage_ms"]["describe"], "%.1f%%" % (100*(v["value"]/d["value"]-1)))
print("cpu %.3g (%.3g)" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["single_thread"]["value"]), d["cpu_baseline"]["cores"])
for k,e in d["roofline"]["stages"].items(): print(k, e.get("limiter"), {a:round(b,3) for a,b in (e.get("limiter_utilisation") or {}).items()}, "traffic %.3g alg %.3g" % (e.get("traffic") or 0, e.get("algorithmic_bytes") or 0))
i=d["ingest"]; print("ingest %.4g kp/s %.0f fps eff %.3f %s" % (i["value"], i["frames_per_s"], i["overlap_efficiency"], i["bound_by"]), i["ms_per_slot"], i["h2d_GBps"])
sp=d["scaling_prediction"]; print("pred", sp["auto_form"], sp["fixed_stride"]["at_link_peak"], sp["fixed_stride"]["at_0.8_of_link_peak"])
a=json.load(open('profiles/r05_final_align_bench.json'))
print("align %.0f fps %.4f ms roof %.4f ach %.0f traffic %.3g px/s %.4g cpu %.0f (%.0f)" % (a["value"], a["ms_per_step"], a["roofline"]["frac"], a["roofline"]["achieved"], a["roofline"]["traffic"] or 0, a["pixels_per_s"], a["cpu_baseline"]["value"], a["cpu_baseline"]["single_thread"]["value"]), a["roofline"]["pmc"])
for m in ("ref","c3","c4","c5"):
    x=json.load(open('profiles/r05_final_%s_bench.json'%m)); print(m, "%.4g"%x["value"], round(x["ms_per_step"],4), "%.4g fps"%x["frames_per_s"], "cpu %.3g"%(x["cpu_baseline"] or {}).get("value",0), "path %.3f %.0f" % (x["path_hbm"]["frac_of_8TBps"], x["path_hbm"]["achieved_GBps"]), x.get("matcher_gpairs_per_s"), "stale", x["roofline"]["pmc"]["stale"], {k:round(v,4) for k,v in x["stage_ms"].items()})
for name in ("match","match_stream","match_tile"):
    x=json.load(open('profiles/r05_final_%s_bench.json'%name))
    for s in x["sizes"]: print(name, s["n"], s["brute_force_256bit"]["kernels"], round(s["brute_force_256bit"]["ms_per_call"],4), round(s["brute_force_256bit"].get("gpairs_per_s",0),0), round(s["reference_32bit_window2"]["ms_per_call"]*1e3,1))
g=json.load(open('profiles/r05_final_ingest_bench.json'))
for r in g["runs"]: print("ingest", r["frames_per_slot"], r["input"], r["source"][:8], "%.0f fps"%r["frames_per_s"], "eff %.3f"%r["overlap_efficiency"], r["bound_by"], "h2d %.1f of %.1f"%(r["h2d_GBps"]["achieved_pipelined"], r["h2d_GBps"]["peak_measured_pinned"]), {k:round(v,3) for k,v in r["ms_per_slot"].items()})
PY
awk -F'",' 'NR>1 && NR<8 {print substr($1,1,40), $2}' profiles/r05_final_kernel_stats.csv; awk -F'",' 'NR>1 && NR<5 {print substr($1,1,40), $2}' profiles/r05_final_align_kernel_stats.csv
grep -v amdgpu profiles/r05_final_stage_latency.txt profiles/r05_final_latency_probe.txt | cut -d: -f2-
