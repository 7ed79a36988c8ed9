"""Markdown rows of DESIGN.md section 5 / README numbers from a bench_detail.json (python tools/rounds/r06_doc_numbers.py <file>)."""
import json
import sys

d = json.load(open(sys.argv[1]))


def M(x):
    return f"{x / 1e6:.2f} M"


k = {n: v["tflops"] for n, v in d["kernels"].items()}
r = d["roofline"]
a = d["amp_full"]
ak = {n: (v["ms_total"] / a["steps"], v["tflops"]) for n, v in a["kernels"].items()}
ast = {n: v for n, v in a["stages"].items()}
fo, zs = d["forward_only"], d["zero_shot"]
go = "GO-2019 (32102 labels x 2 descriptions)"
ec = "EC (5134 labels x 2 descriptions)"
c = d.get("cpu_baseline") or {}
rows = [
    f"| **headline: `configs[2]` train step** (fwd + bwd + clip + Adam) | f32 | **{d['ms_per_step']:.0f}** | **{M(d['value'])}** | **{r['achieved']:.1f} TFLOP/s = {r['frac']:.3f} of 157.3** ({r['launches']} launches in {d['steps']} steps, {r['avg_ms_per_launch']:.1f} ms mean; family = {100 * r['family_share_of_step']:.1f} % of the step); `traffic` {r['traffic'] / 1e9:.0f} GB per full-grid launch = {r['traffic'] / 202e9:.2f} x algorithmic (`traffic_stale: {str(r['traffic_stale']).lower()}`) |",
    f"| same, per kind | f32 | - | - | `nt:plain` {k['nt:plain']}, `nt:bn_relu` {k['nt:bn_relu(z)']}, `nt:pairsum` {k['nt:pairsum_relu']}, `tn:bn_relu` {k['tn:plain x bn_relu']}, `tn:pairsum` {k['tn:plain x pairsum']}, encoder `nt:conv` {k['nt:conv']} TFLOP/s |",
    f"| same, dense definition of SURVEY §8d (151.0 MFLOP per pair incl. the eliminated layer-1 work) | - | - | - | {r['whole_step_tflops_dense_definition']:.1f} TFLOP/s \"dense\", {r['whole_step_tflops_issued']:.1f} TFLOP/s issued over the whole step |",
    f"| `fast_mode`: same step, bf16x3 | bf16x3 | {d['fast_mode']['ms_per_step']:.0f} | {M(d['fast_mode']['value'])} | {d['fast_mode']['roofline']['achieved']:.1f} TFLOP/s f32-equivalent = {d['fast_mode']['roofline']['frac']:.3f} of 2.5 PF (ceiling 1/3) |",
]
for key, lab in (("forward_bf16x3", "bf16x3 forward + bf16 backward"), ("forward_f32", "f32 forward + bf16 backward")):
    v = d["amp_backward"][key]
    rows.append(f"| `amp_backward`, {lab} | {lab.split(' ')[0]} / bf16 | {v['ms_per_step']:.0f} | {M(v['value'])} | backward family {v['roofline']['achieved']:.0f} TFLOP/s = {v['roofline']['frac']:.3f} of 2.5 PF |")
fwd = [x for n, x in ak.items() if "operand materialised" in n]
rows.append(
    f"| **`amp_full`** (round 6): bf16x3 base + `forward_math` = `backward_math` = bf16 | bf16 x bf16 -> f32 GEMMs | **{a['ms_per_step']:.0f}** | **{M(a['value'])}** | all six full-grid launches {a['roofline']['achieved']:.0f} TFLOP/s = {a['roofline']['frac']:.3f} of 2.5 PF; forward launches {a['forward_roofline']['achieved']:.0f} = {a['forward_roofline']['frac']:.3f} ({' / '.join(f'{m:.0f} ms' for m, _ in fwd)} per layer pass in 32 chunk launches) + `k_make_h_bf16` {ast['forward operand materialised as bf16 (k_make_h_bf16)']['ms_per_step']:.0f} ms per step at {ast['forward operand materialised as bf16 (k_make_h_bf16)']['achieved_TBps']:.1f} TB/s |")
for key in ("frozen_output_layer", "ragged_lengths"):
    v = d[key]
    rows.append(f"| `{key}` | f32 | {v['ms_per_step']:.0f} | {M(v['value'])} | {v['roofline']['frac']:.3f} |")
rows.append("| `forward_only` = `configs[1]` | f32 / bf16x3 / bf16 (round 6) | " + " / ".join(f"{fo[m]['ms_per_forward']:.0f}" for m in ("f32", "bf16x3", "bf16")) + " | " +
            " / ".join(M(fo[m]["value"]) for m in ("f32", "bf16x3", "bf16")) + " | " + " / ".join(f"{fo[m]['roofline']['frac']:.3f}" for m in ("f32", "bf16x3", "bf16")) + " |")
for m in ("f32", "bf16x3", "bf16"):
    g, e = zs[m][go], zs[m][ec]
    nc = f" ({M(g['value_without_label_projection_cache'])} / {M(e['value_without_label_projection_cache'])} with `L_e` recomputed per batch as the reference does)" if g.get("value_without_label_projection_cache") and e.get("value_without_label_projection_cache") else ""
    rows.append(f"| `zero_shot` = `configs[4]` (512 sequences, L <= 2048 bucketed, 2 descriptions, GO then EC table by name) | {m} | {g['seconds'] * 1e3:.0f} (GO) / {e['seconds'] * 1e3:.0f} (EC) per pass | {M(g['value'])} / {M(e['value'])}{nc} | {g['roofline']['frac']:.3f} / {e['roofline']['frac']:.3f} |")
sh, oh = d["similarity_head"], d["one_hidden_layer"]
rows.append(f"| `similarity_head` train / eval | f32 | {sh['train']['ms_per_step']:.1f} / {sh['eval']['ms_per_forward']:.1f} | {M(sh['train']['value'])} / {M(sh['eval']['value'])} | head contractions {sh['train']['roofline']['frac']:.2f} / {sh['eval']['roofline']['frac']:.2f} of the f32 peak (encoder- and `W_l`-bound step) |")
rows.append(f"| `one_hidden_layer` (`OUTPUT_MLP_NUM_LAYERS: 1`) train / eval | f32 | {oh['train']['ms_per_step']:.1f} / {oh['eval']['ms_per_forward']:.1f} | {M(oh['train']['value'])} / {M(oh['eval']['value'])} | VALU-bound head: {oh['train']['roofline']['frac']:.2f} / {oh['eval']['roofline']['frac']:.2f} of the rate its instruction mix can reach (forward add + max + fma: 59 T operations/s) |")
if c.get("value"):
    legs = "; ".join(f"{n}: {v['value'] / 1e3:.2f} k ({v['seconds']:.1f} s)" for n, v in c["legs"].items() if v.get("value"))
    rows.append(f"| `cpu_baseline` (oracle port, L = 512, 32 102 labels, full width) | f32 | {c['legs'][str(c['cores']) + ' (B=4)']['seconds'] * 1e3 if str(c['cores']) + ' (B=4)' in c['legs'] else 0:.0f} | {c['value'] / 1e3:.2f} k on {c['cores']} threads (host: {c['host_cores']} hardware threads, cgroup quota {c.get('usable_cpus')} CPUs); legs (threads: pairs/s, B = 2 unless marked) {legs} | - |")
print("\n".join(rows))
st = d["stages"]
print("\nstages:", "; ".join(f"{n.split('(')[-1].rstrip(')') if '(' in n else n} {v['ms_per_step']:.1f} ms at {v.get('achieved_TBps')} TB/s" for n, v in st.items()))
print("bench wall:", d.get("build_hash"))
