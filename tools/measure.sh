#!/bin/bash
TAG=${1:-r03}
export ROUND_TAG=$TAG
# The round's measurement pass on one MI355X: PMC passes for every bench configuration -> profiles/${TAG}_pmc.json (keyed by the hash of
# the kernel sources), bench.py plain -> ${TAG}_bench.json, bench.py under rocprofv3 --kernel-trace --stats -> kernel stats CSV.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
rm -rf gpurun_out/$TAG/trace
if [ -z "$SKIP_PMC" ]; then   # (the two halves fit one 20-minute gpurun call each: SKIP_PMC=1 runs the second half only)
rm -rf gpurun_out/pmc2
bash tools/prof_pmc2.sh > gpurun_out/$TAG/pmc2.log 2>&1 || { tail -20 gpurun_out/$TAG/pmc2.log; exit 1; }
tail -7 gpurun_out/$TAG/pmc2.log
cp gpurun_out/${TAG}_pmc.json profiles/${TAG}_pmc.json
fi
timeout -k 10 600 python bench.py > gpurun_out/$TAG/${TAG}_bench.json 2> gpurun_out/$TAG/${TAG}_bench.err || { tail -5 gpurun_out/$TAG/${TAG}_bench.err; exit 1; }
python3 - <<'PY'
import json, os
T=os.environ['ROUND_TAG']
b=json.load(open(f"gpurun_out/{T}/{T}_bench.json"))
print("value", b["value"], "ms/step", b["ms_per_step"])
r=b["roofline"]; print("roofline eff frac", r["frac"], "valu", r["valu_issue"], "lds", r["lds"], "hbm", r["hbm_physical"])
for k,c in b.get("configs",{}).items():
    if "error" in c: print(k, c["error"]); continue
    print(k, "worst %.3f M fr/s"%(c["worst_case"]["value"]/1e6), "oper %.3f M"%(c["operating_point"]["value"]/1e6), "kernel_ms", round(c["roofline"]["kernel_ms_avg"],3), "valu frac", (c["roofline"]["valu_issue"] or {}).get("frac"), "hbm frac", (c["roofline"]["hbm_physical"] or {}).get("frac"))
print("cpu", b["cpu_baseline"])
PY
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace -- python3 bench.py --no-cpu-baseline > gpurun_out/$TAG/${TAG}_bench_under_rocprof.json 2> gpurun_out/$TAG/trace.err || { tail -5 gpurun_out/$TAG/trace.err; exit 1; }
# bench.py starts child processes (abi_multi, exact_replay_ranks), each with its own trace: the main process's is the longest
ls -S gpurun_out/$TAG/trace/*/*kernel_trace.csv | head -1 | sed 's/kernel_trace/kernel_stats/' | xargs -I{} cp {} gpurun_out/$TAG/${TAG}_kernel_stats.csv
ls -S gpurun_out/$TAG/trace/*/*kernel_trace.csv | head -1 | xargs -I{} cp {} gpurun_out/$TAG/${TAG}_kernel_trace.csv
python3 tools/trace_legs.py gpurun_out/$TAG/${TAG}_kernel_trace.csv gpurun_out/$TAG/${TAG}_bench_under_rocprof.json > gpurun_out/$TAG/${TAG}_kernel_trace_legs.txt 2>&1 || true
head -12 gpurun_out/$TAG/${TAG}_kernel_stats.csv | cut -c1-160
# the generation shared out over logical shards (one GPU: the device does the work once whatever n)
timeout -k 10 300 python tools/time_shards.py 65536 > gpurun_out/$TAG/${TAG}_exact_replay_shards.json 2> gpurun_out/$TAG/time_shards.err || { tail -5 gpurun_out/$TAG/time_shards.err; exit 1; }
# the exact-replay path: generator, noise -> decode -> count, harness end to end, whole-call latency, code-search-like loop
timeout -k 10 500 python tools/time_exact.py > gpurun_out/$TAG/${TAG}_exact_replay.json 2> gpurun_out/$TAG/time_exact.err || { tail -5 gpurun_out/$TAG/time_exact.err; exit 1; }
python3 - <<'PY'
import json, os
T=os.environ['ROUND_TAG']
d=json.load(open(f"gpurun_out/{T}/{T}_exact_replay.json"))
for k in ("generator","mt_frames_B65536","decode_only_B65536","harness_device_noise","harness_device_long_run_noise","harness_host_noise"): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in d[k].items()})
for k,v in d["whole_call_latency_best_of_3"].items(): print(k, round(v["device"]["seconds"]*1e3,2), "ms device", round(v["host"]["seconds"]*1e3,2), "ms host")
for k,v in d["code_search_like_loop_tasp_m64"].items(): print("search loop", k, "mean s/candidate", round(v["mean_seconds"],3), [round(p["seconds"],3) for p in v["candidates"]])
PY
rm -rf gpurun_out/$TAG/trace_exact
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace_exact -- python3 tools/time_exact.py --frames 100000 --host-frames 100 > /dev/null 2> gpurun_out/$TAG/trace_exact.err || { tail -5 gpurun_out/$TAG/trace_exact.err; exit 1; }
find gpurun_out/$TAG/trace_exact -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/$TAG/${TAG}_exact_replay_kernel_stats.csv
head -9 gpurun_out/$TAG/${TAG}_exact_replay_kernel_stats.csv | cut -c1-150
