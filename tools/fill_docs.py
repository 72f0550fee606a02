"""Fills the @NAME@ placeholders of the doc templates (DESIGN.md / README.md kept under a template directory) from the committed
bench lines and profile summaries of a round: python tools/fill_docs.py <template dir> <round tag, e.g. r04>"""
import csv, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tpl, tag = sys.argv[1], sys.argv[2]
P = lambda n: os.path.join(ROOT, "profiles", n)
L = lambda n: json.load(open(P("bench_%s_%s.json" % (tag, n))))
d, nov, nol, c5 = L("default"), L("nooverlap"), L("nolazy"), L("C5")
fo = d["dropin"]["fdgs_optim"]
rf = d["roofline"]
# VALU issue fraction from the committed SQ pass of THIS round (the bench line of the same run still read the previous file)
cur, sq = None, {}
for line in open(P("pmc_sq_%s.txt" % tag)):
    t = line.split()
    if not line.startswith(" ") and t: cur = line.strip()
    elif cur and "blend_bwd_kernel<false>" in cur and len(t) >= 2: sq[t[0]] = float(t[1])
valu = sq["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (sq["_AVG_DURATION_NS"] * 1e-9 * rf["shader_clock_ghz_measured"] * 1e9)
bbr = None
for r in csv.DictReader(open(P("kernel_stats_%s.csv" % tag))):
    if "blend_bwd_kernel<false>" in r["Name"]: bbr = float(r["AverageNs"]) / 1e6
tl = open(P("step_timeline_%s.txt" % tag)).read()
g = lambda pat: re.search(pat, tl).group(1)
v = {
    "STEP": "%.0f" % d["value"], "MS": "%.2f" % d["ms_per_step"], "MSI": "%.2f" % d["ms_per_image"], "MED": "%.2f" % d["ms_per_step_median"],
    "P10": "%.2f" % d["ms_per_step_p10"], "P90": "%.2f" % d["ms_per_step_p90"], "NSTEPS": str(d["steps_timed"]),
    "REFL": "%.0f" % d["value_reference_lists"], "RAND": "%.0f" % d["random_order_images_s"], "CLU": "%.0f" % d["clustered_images_s"],
    "NOOV": "%.0f" % nov["value"], "NOLAZY": "%.0f" % nol["value"], "FWD": "%.3f" % d["forward_ms"], "MPIX": "%.0f" % d["forward_mpix_s"],
    "RASTER": "%.0f" % d["raster_images_s"], "CPU": "%.2f" % d["cpu_baseline"]["value"],
    "D0": "%.0f" % d["dropin"]["images_s"], "D1": "%.0f" % d["dropin"]["images_s_with_fused_loss"], "D2": "%.0f" % fo["images_s_fdgs_optim"],
    "D3": "%.0f" % fo["images_s_fdgs_optim_lazy"], "D4": "%.0f" % fo["images_s_fdgs_optim_lazy_tile_cull"], "DF": "%.2f" % fo["forward_ms_fdgs_optim"],
    "BB": "%.3f" % rf["avg_kernel_ms"], "BBS": "%.3f" % rf["avg_kernel_ms_single_stream"], "BBR": "%.3f" % bbr,
    "BBG": "%.2f" % (rf["achieved"] / 1e3), "BBF": "%.3f" % rf["frac"], "VALU": "%.2f" % valu, "VINS": "%.0f" % (sq["SQ_INSTS_VALU"] / 1e6),
    "C5": "%.0f" % c5["value"], "C5F": "%.3f" % c5["forward_ms"], "C5M": "%.0f" % c5["forward_mpix_s"],
    "TLV": g(r"valu-bound kernel running\s+([\d.]+) %"), "TLL": g(r"only latency/HBM-bound kernels\s+([\d.]+) %"), "TLI": g(r"idle\s+([\d.]+) %"),
    "TLS": g(r"sum of kernel durations / wall = ([\d.]+)"),
}
for name in ("DESIGN.md", "README.md"):
    s = open(os.path.join(tpl, name)).read()
    for k, val in v.items():
        s = s.replace("@%s@" % k, val)
    left = re.findall(r"@[A-Z0-9]+@", s)
    assert not left, left
    open(os.path.join(ROOT, name), "w").write(s)
print(json.dumps(v))
