#!/usr/bin/env python3
"""Rewrites DESIGN.md section 9 ("Numbers of this round") from the committed measurement files of a round, so that the document quotes
exactly what is under profiles/:   python tools/design_table.py [r03]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"


def load(name):
    p = os.path.join(ROOT, "profiles", f"{TAG}_{name}")
    return json.load(open(p)) if os.path.exists(p) else None


b = load("bench.json")
labels = {"cfg3_sum_product": "cfg3 (2048,1024) sum-product 50 it", "cfg4_layered_m512": "cfg4 (16384,8192) layered min-sum 50 it",
          "cfg5_qam16_min_sum": "cfg5 (2048,1024) min-sum behind 16-QAM", "f1_integer_min_sum": "f1 (2048,1024) integer min-sum 50 it",
          "f2_tasp_m126": "f2 (4032,2016) TDMP sum-product 15 it (shipped scenario)", "f2_bp_m64": "f2 (2048,1024) Gallager BP 50 it",
          "f2_asp_m64": "f2 (2048,1024) flooding sum-product (prob.) 50 it"}


def row(label, kern, worst, oper, rf):
    v, l, h = rf.get("valu_issue") or {}, rf.get("lds") or {}, rf.get("hbm_physical") or {}
    eff = f"{rf['frac']:.2f}" if rf.get("frac") else "—"
    conf = f"{l['bank_conflict_cycles'] / 1e6:.1f} M" if l.get("bank_conflict_cycles") is not None else "—"
    return (f"| {label} | `{kern.split('(')[0].strip()}` | {worst['value'] / 1e6:.3f} M | {oper['value'] / 1e6:.3f} M @ {oper['ebn0_db']:.1f} dB "
            f"({oper['mean_iters_per_frame']:.1f} it, FER {oper['fer']:.4f}) | {rf['kernel_ms_avg']:.2f} | {eff} | "
            f"{v.get('frac', 0):.2f} ({v.get('valu_insts_per_wave_iter', 0):.0f}) | {l.get('frac', 0):.2f} | {conf} | {h.get('frac', 0):.3f} |")


out = ["## 9. Numbers of this round", "",
       f"From `profiles/{TAG}_bench.json` (one `python bench.py` run on an MI355X; PMC figures from `profiles/{TAG}_pmc.json`, same sources). "
       "Worst case = Eb/N0 0 dB, every frame runs all iterations, LLRs resident in HBM.", "",
       "| configuration | kernel | frames/s worst case | frames/s at the operating point | kernel ms / launch | effective §8(d) fraction of 8 TB/s | "
       "VALU-issue fraction (instr. per wave-iteration) | LDS busy | LDS bank-conflict cycles / launch | physical HBM fraction |",
       "|---|---|---|---|---|---|---|---|---|---|"]
if b:
    op = dict(b["operating_point"])
    out.append(row("cfg2 (2048,1024) min-sum 50 it — headline", b["roofline"]["kernel"], {"value": b["value"]}, op, b["roofline"]))
    for k, c in b.get("configs", {}).items():
        if "error" in c:
            out.append(f"| {labels.get(k, k)} | error: {c['error'][:80]} | | | | | | | | |")
            continue
        out.append(row(labels.get(k, k), c["roofline"]["kernel"], c["worst_case"], c["operating_point"], c["roofline"]))
    o1 = b["operating_point"].get("one_batch_at_a_time", {})
    out += ["",
            f"Headline: **{b['value'] / 1e6:.3f} M frames/s** ({b['ms_per_step']:.2f} ms per 65 536-frame step, mean {b['config']['mean_iters_per_frame']:.2f} iterations). "
            f"Operating point 2.0 dB: {b['operating_point']['value'] / 1e6:.2f} M frames/s with two batches in flight "
            f"({100 * b['operating_point']['fraction_of_worst_case_frame_iterations_per_s']:.1f} % of the worst case's frame-iterations/s), "
            f"{o1.get('value', 0) / 1e6:.2f} M one batch at a time ({100 * o1.get('fraction_of_worst_case_frame_iterations_per_s', 0):.1f} %)."]
    cpu = b.get("cpu_baseline") or {}
    if cpu.get("value"):
        out.append(f"CPU reference on the same box (compiled upstream `min_sum_decod_qc_lm`, same workload, decode only): {cpu['value_1core']:.0f} frames/s on one core, "
                   f"{cpu['value']:.0f} frames/s on {cpu['cores']} cores.")
    am = b.get("abi_multi") or {}
    if am.get("value"):
        out.append(f"`abi_multi` (one shard through `ldpc_hip_decode_count_multi`): {am['value'] / 1e6:.3f} M frames/s, {am['ms_per_step']:.2f} ms per step.")
    er = b.get("exact_replay") or {}
    if er.get("value"):
        g = er["generator"]
        out += ["",
                f"Exact replay (`exact_replay` of the bench line): first 4001 frames {er['headline_run']['errored_frames']} errored (upstream binary: 170); "
                f"**{er['value'] / 1e6:.2f} M frames/s** noise → decode → count on 65 536-frame batches at 2.0 dB; generator alone "
                f"**{g['samples_per_s'] / 1e9:.1f} G samples/s** = {g['achieved_GBs'] / 1e3:.2f} TB/s of its {g['algorithmic_bytes_per_sample']:.1f} B/sample algorithmic traffic "
                f"= {g['frac']:.2f} of the HBM roof" + (f" (PMC: {g['traffic'] / g['samples']:.1f} B/sample physical)." if g.get("traffic") else ".")]
    rk = b.get("exact_replay_ranks") or {}
    if rk.get("value"):
        out.append(f"`exact_replay_ranks` (`host.bp_simulation(exact_seed=1)` as a one-process-per-GPU job, here {rk['n_gpus']} rank; the N > 1 lines must repeat these "
                   f"counters): {rk['value'] / 1e6:.2f} M frames/s, {rk['ms_per_round']:.2f} ms per {rk['frames_per_round']}-frame round, {rk['errored_frames']} errored of "
                   f"{rk['frames']} frames, generator state CRC {rk['generator_state_crc32']:#010x}.")
sh = load("exact_replay_shards.json")
if sh:
    out += ["", "Generation shared out over n logical shards on ONE GPU (`tools/time_shards.py`, ms per 65 536 frames; on one device the work "
            "is done once whatever n when the tape is shared out, n times when every shard makes the whole tape):", "",
            "| n | tape shared out: generate only | noise → decode → count | whole tape per shard: generate only | noise → decode → count |", "|---|---|---|---|---|"]
    for n in ("1", "2", "4", "8"):
        a_, w_ = sh["tape_shared_out"][n], sh["whole_tape_on_every_shard"][n]
        out.append(f"| {n} | {a_['generate_only_ms']:.2f} | {a_['noise_decode_count_ms']:.2f} | {w_['generate_only_ms']:.2f} | {w_['noise_decode_count_ms']:.2f} |")
ex = load("exact_replay.json")
if ex:
    h = ex.get("harness_device_long_run_noise", {})
    out += ["", f"C++ harness end to end (`tools/time_exact.py`): {h.get('frames', 0):,} frames in {h.get('seconds', 0):.2f} s = {h.get('frames_per_s', 0) / 1e6:.2f} M frames/s "
            f"(host noise: {ex.get('harness_host_noise', {}).get('frames_per_s', 0) / 1e3:.1f} k frames/s)."]
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a = s.index("## 9. Numbers of this round")
s = s[:a] + "\n".join(out) + "\n"
open(p, "w").write(s)
print("\n".join(out))
