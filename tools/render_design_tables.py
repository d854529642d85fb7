#!/usr/bin/env python3
"""Generated tables of DESIGN.md / README.md: the parity table and the figure table are RENDERED from the tracked evidence files of the
latest round -- profiles/rNN_parity_fulldepth.json (tests/test_gpu_fulldepth.py), profiles/rNN_parity_batch_fulldepth.json
(tests/test_gpu_batch_fulldepth.py), profiles/rNN_bench_line_final.json (bench.py) -- between marker comments

    <!-- generated:<name> begin (tools/render_design_tables.py) -->  ...  <!-- generated:<name> end -->

so that a number in those tables cannot drift from the measurement (round-5 review: the parity table quoted a kernel variant that was
not the shipped default).  tests/test_docs_generated.py re-renders and fails when a document differs.
usage: render_design_tables.py [--check]      (no flag: rewrite the blocks in place)"""
import glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ("DESIGN.md", "README.md")


def latest(pattern):
    """the evidence file of the highest round that has one"""
    best = None
    for p in glob.glob(os.path.join(ROOT, "profiles", pattern)):
        m = re.match(r"r(\d+)_", os.path.basename(p))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), p)
    if best is None:
        raise FileNotFoundError(pattern)
    return best[1]


def load():
    files = dict(single=latest("r*_parity_fulldepth.json"), batch=latest("r*_parity_batch_fulldepth.json"), bench=latest("r*_bench_line_final.json"))
    return {k: json.load(open(v)) for k, v in files.items()}, {k: os.path.relpath(v, ROOT) for k, v in files.items()}


def per_lane(entry, n_cases=2):
    """'375 / 242': the counts of the lanes that decode golden utterance 0 / 1 (lanes that decode the same utterance agree)"""
    pl = entry["per_lane"]
    return " / ".join(str(pl[i]) for i in range(min(n_cases, len(pl))))


def parity_table(ev, src):
    s, b = ev["single"], ev["batch"]
    rows = ["| model | path | dtype | identical decisions (of 384; batch: per lane, two golden utterances of 384 / 256 decisions) | worst mismatch (oracle top-2 margin, bf16 ulps) | unexplained |",
            "|---|---|---|---|---|---|"]
    for size, name in (("0p6b", "0.6B"), ("1p7b", "1.7B")):
        for dt in ("f32", "bf16"):
            e = s.get(f"{size}_{dt}")
            if e:
                rows.append(f"| {name} (28 + 5 layers) | single stream | {'fp32' if dt == 'f32' else 'bf16'} | {e['matched_decisions']} / {e['total']} "
                            f"({e['matched_frames']} / {e['frames']} whole frames) | {e['worst_mismatch_ulp']:g} | {e['unexplained']} |")
    for size, name in (("0p6b", "0.6B"), ("1p7b", "1.7B")):
        for B in (8, 16, 32, 64, 128):
            e = b.get(f"{size}_bf16_mfma_B{B}")
            if e:
                form = "panel MFMA GEMVs" if B <= 32 else "weight-stationary GEMMs on fragment-major weights, lane attention, pair pass (the default)"
                rows.append(f"| {name} | {B} lanes, {form} | bf16 | {per_lane(e)} | {e['worst_mismatch_ulp']:g} | {e['unexplained']} |")
        e = b.get(f"{size}_bf16_mfma_B64_round4_form")
        if e:
            rows.append(f"| {name} | 64 lanes, the round-4 form (`attn_lane` 0) | bf16 | {per_lane(e)} | | |")
    e = b.get("0p6b_f32_valu_B32")
    if e:
        rows.append(f"| 0.6B | 32 lanes, VALU GEMVs | fp32 | {e['matched']} / {e['total']} | {e.get('worst_mismatch_ulp', 0):g} | {e.get('unexplained', 0)} |")
    c4 = (ev["bench"].get("model_1p7b") or {}).get("config4_voice_design_4k", {}).get("parity")
    if isinstance(c4, dict) and "matched_decisions" in c4:
        rows.append(f"| 1.7B, 4096-token prompt (configs[4]) | single stream | bf16 | {c4['matched_decisions']} / {c4.get('decisions', 128)} | "
                    f"{c4.get('worst_mismatch_margin_bf16_ulp', '')} | {c4.get('unexplained', '')} |")
    rows.append("")
    rows.append(f"(rendered from `{src['single']}`, `{src['batch']}`, `{src['bench']}`)")
    return "\n".join(rows)


def fig(x, nd=1):
    return "n/a" if x is None else (f"{x:.{nd}f}" if isinstance(x, (int, float)) else str(x))


def figures_table(ev, src):
    d = ev["bench"]
    b = d.get("batched_decode_one_gpu", {}) or {}
    m = d.get("model_1p7b", {}) or {}
    r, br = d.get("roofline", {}) or {}, b.get("roofline", {}) or {}
    st = lambda k: b.get(k) or {}
    rm = d.get("roofline_mfma", {}) or {}
    c4 = m.get("config4_voice_design_4k", {}) or {}
    pp = d.get("parity_pcm", {}) or {}
    pf = d.get("parity_bf16_frames", {}) or {}
    cpu = d.get("cpu_baseline", {}) or {}
    rows = ["| figure | value |", "|---|---|",
            f"| RTF, single stream, end to end (`value`) | **{fig(d.get('value'), 2)}x** -- target >= 30 |",
            f"| p50 TTFA | **{fig(d.get('ttfa_ms_p50'))} ms** -- target < 150 |",
            f"| decode frame graph | {fig(d.get('decode_ms_per_frame'), 3)} ms; `roofline.frac` **{fig(r.get('frac'), 4)}**; `traffic` {fig((r.get('traffic') or 0) / 1e9, 3)} GB = "
            f"{fig(r.get('traffic_over_algorithmic'), 2)} x algorithmic |",
            f"| {b.get('lanes', 128)} lock-step lanes (`batched_decode_one_gpu`) | **{fig(b.get('ms_per_lockstep_frame'), 3)} ms per frame = {fig(b.get('decode_only_value'), 0)}x decode only**; "
            f"**{fig(b.get('value'), 0)}x end to end** ({fig(b.get('end_to_end_over_decode_only'), 3)} of decode only); {fig(st('streaming').get('value'), 0)}x streamed; "
            f"`roofline.frac` **{fig(br.get('frac'), 4)}**; `traffic` {fig((br.get('traffic') or 0) / 1e9, 2)} GB = {fig(br.get('traffic_over_algorithmic'), 2)} x algorithmic |",
            f"| first audio of N simultaneous streaming requests (p50 / max) | 128: {fig(st('streaming').get('ttfa_ms_first_wave_p50'))} / {fig(st('streaming').get('ttfa_ms_first_wave_max'))} ms; "
            f"64: {fig(st('streaming_64_lanes').get('ttfa_ms_first_wave_p50'))} / {fig(st('streaming_64_lanes').get('ttfa_ms_first_wave_max'))} ms; "
            f"32: {fig(st('streaming_32_lanes').get('ttfa_ms_first_wave_p50'))} / {fig(st('streaming_32_lanes').get('ttfa_ms_first_wave_max'))} ms |",
            f"| 64 / 32 / 16 lanes, decode only | {fig(st('lanes_64').get('ms_per_lockstep_frame'), 3)} ms = {fig(st('lanes_64').get('decode_only_value'), 0)}x; "
            f"{fig(st('lanes_32').get('ms_per_lockstep_frame'), 3)} ms = {fig(st('lanes_32').get('decode_only_value'), 0)}x; "
            f"{fig(st('lanes_16').get('ms_per_lockstep_frame'), 3)} ms = {fig(st('lanes_16').get('decode_only_value'), 0)}x |",
            f"| configs[3] (`config3_sharded_batched`) | {fig((d.get('config3_sharded_batched') or {}).get('wall_s'), 3)} s = **{fig((d.get('config3_sharded_batched') or {}).get('value'), 0)}x** |",
            f"| configs[2] (`model_1p7b`) | {fig(m.get('rtf'), 2)}x RTF, TTFA {fig(m.get('ttfa_ms_p50'))} ms, {fig(m.get('decode_ms_per_frame'), 3)} ms / frame "
            f"(`frac` {fig((m.get('roofline') or {}).get('frac'), 3)}); lock-step 32 / 64 / 128 lanes {fig((m.get('batched_b32') or {}).get('ms_per_lockstep_frame'), 3)} / "
            f"{fig((m.get('batched_b64') or {}).get('ms_per_lockstep_frame'), 3)} / {fig((m.get('batched_b128') or {}).get('ms_per_lockstep_frame'), 3)} ms = "
            f"{fig((m.get('batched_b32') or {}).get('value'), 0)} / {fig((m.get('batched_b64') or {}).get('value'), 0)} / {fig((m.get('batched_b128') or {}).get('value'), 0)}x |",
            f"| configs[4] (`config4_voice_design_4k`) | {c4.get('prompt_rows', 4096)}-row prompt: {fig(c4.get('rtf'), 2)}x RTF, TTFA {fig(c4.get('ttfa_ms_p50'))} ms |",
            f"| 4 single-stream utterances in flight | {fig((d.get('concurrent_utterances_one_gpu') or {}).get('value'), 1)}x aggregate, TTFA p50 {fig((d.get('concurrent_utterances_one_gpu') or {}).get('ttfa_ms_p50'))} ms |",
            f"| MFMA rooflines (`roofline_mfma`) | codec 370 frames {fig((rm.get('codec_full_decode') or {}).get('ms'), 2)} ms = {fig((rm.get('codec_full_decode') or {}).get('achieved'), 0)} TFLOP/s "
            f"({fig(100 * ((rm.get('codec_full_decode') or {}).get('frac') or 0), 1)} %); prefill-200 {fig((rm.get('prefill_200') or {}).get('ms'), 3)} ms = "
            f"{fig((rm.get('prefill_200') or {}).get('achieved'), 0)} TFLOP/s ({fig(100 * ((rm.get('prefill_200') or {}).get('frac') or 0), 1)} %); prefill-4096 "
            f"{fig((m.get('prefill_4096') or {}).get('ms'), 2)} ms = {fig((m.get('prefill_4096') or {}).get('achieved'), 0)} TFLOP/s ({fig(100 * ((m.get('prefill_4096') or {}).get('frac') or 0), 1)} %)"
            + "".join(f"; {k.replace('_', ' ')} {fig(v.get('ms'), 2)} ms = {fig(v.get('ms_per_prompt'), 3)} ms per prompt ({fig(100 * (v.get('frac') or 0), 1)} %)"
                      for k, v in sorted(rm.items()) if k.startswith('packed_prefill_') and isinstance(v, dict) and 'ms' in v) + " |",
            f"| `parity_bf16_frames` | {pf.get('matched_decisions')} / {pf.get('decisions')} decisions, {pf.get('matched_frames')} / {pf.get('frames')} whole frames, worst "
            f"{fig(pf.get('worst_mismatch_margin_bf16_ulp'), 0)} ulps, unexplained {pf.get('unexplained')} |",
            f"| `parity_pcm` (vs the fp32-arithmetic oracle) | bf16x2 {fig((pp.get('bf16x2') or {}).get('pcm_rms_vs_fp32_oracle'), 7)}; bf16 {fig((pp.get('bf16') or {}).get('pcm_rms_vs_fp32_oracle'), 5)}; "
            f"fp32 {fig((pp.get('fp32') or {}).get('pcm_rms_vs_fp32_oracle'), 8)} |",
            f"| vocoder (`bf16x2`) | full 370 frames {fig((pp.get('bf16x2') or {}).get('full_decode_370_frames_ms'), 2)} ms; streaming chunk {fig((pp.get('bf16x2') or {}).get('streaming_chunk_8_frames_ms'), 2)} ms; "
            f"16 utterances batched, tail after the reference {fig((pp.get('bf16x2') or {}).get('batched_16_utterances_tail_after_reference_ms_per_utterance'), 2)} ms per utterance |",
            f"| CPU baseline (`cpu_baseline`, {cpu.get('kind')}, {cpu.get('cores')} cores) | {fig(cpu.get('value'), 3)}x |",
            "", f"(rendered from `{src['bench']}`)"]
    return "\n".join(rows)


BLOCKS = {"parity-table": parity_table, "figures-table": figures_table}


def render(text, ev, src):
    def sub(m):
        name = m.group(1)
        if name not in BLOCKS:
            raise KeyError(f"unknown generated block: {name}")
        return f"<!-- generated:{name} begin (tools/render_design_tables.py) -->\n{BLOCKS[name](ev, src)}\n<!-- generated:{name} end -->"
    return re.sub(r"<!-- generated:([\w-]+) begin[^>]*-->.*?<!-- generated:\1 end -->", sub, text, flags=re.S)


def main():
    check = "--check" in sys.argv
    ev, src = load()
    bad = []
    for doc in DOCS:
        p = os.path.join(ROOT, doc)
        old = open(p).read()
        new = render(old, ev, src)
        if new != old:
            bad.append(doc)
            if not check:
                open(p, "w").write(new)
    if check and bad:
        print("stale generated tables in: " + ", ".join(bad) + " -- run tools/render_design_tables.py")
        return 1
    print(("rewrote: " + ", ".join(bad)) if bad else "generated tables are up to date")
    return 0


if __name__ == "__main__":
    sys.exit(main())
