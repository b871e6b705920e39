#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per conv dispatch: clock, MFMA utilisation, wait shares, HBM bytes.
usage: pmc_report.py <dir with *_counter_collection.csv and *_kernel_trace.csv> [fetch_dir] [write_dir]"""
import csv, glob, sys, collections

def load(d):
    cc = glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv")
    kt = glob.glob(d + "/*/*_kernel_trace.csv") + glob.glob(d + "/*_kernel_trace.csv")
    rows = list(csv.DictReader(open(cc[0])))
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(kt[0]))}
    disp = collections.OrderedDict()
    for r in rows:
        e = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"]),
                                               "vgpr": r["VGPR_Count"], "agpr": r["Accum_VGPR_Count"], "lds": r["LDS_Block_Size"]})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for k, e in disp.items():
        t = trace[k]
        e["dur"] = int(t["End_Timestamp"]) - int(t["Start_Timestamp"])
    return disp

main = load(sys.argv[1])
fetch = load(sys.argv[2]) if len(sys.argv) > 2 else {}
write = load(sys.argv[3]) if len(sys.argv) > 3 else {}
import re
KEY = re.compile(r"(conv3d_\w+_kernel|ww_\w+_kernel|wino_input_kernel|conv1_\w+_kernel|pack_x_bf16_kernel|prologue_\w+_kernel|wgrad_reduce_kernel|splitk_reduce_kernel|attn_\w+|flash\w*)")
fl = [e for e in fetch.values() if KEY.search(e["name"])]
wl = [e for e in write.values() if KEY.search(e["name"])]
i = 0
for k, e in main.items():
    if not KEY.search(e["name"]):
        continue
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", e["name"])
    name = (m.group(1).replace("conv3d_", "").replace("_kernel", "") + (m.group(2) or ""))[:44]
    gui = e.get("GRBM_GUI_ACTIVE", 0); wc = max(e.get("SQ_WAVE_CYCLES", 1), 1)
    s = f"{name:44s} blocks={e['grid'] // e['wg']:5d} v{e['vgpr']}+a{e['agpr']} lds={e['lds']:>6} dur={e['dur'] / 1e3:8.1f}us clk={gui / 8 / e['dur']:4.2f}GHz " \
        f"mfma={e.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (128 * gui) if gui else 0:5.3f} wait_any={e.get('SQ_WAIT_ANY', 0) / wc:4.2f} " \
        f"wait_inst={e.get('SQ_WAIT_INST_ANY', 0) / wc:4.2f} active={e.get('SQ_ACTIVE_INST_ANY', 0) / wc:4.2f} ldsconf={e.get('SQ_LDS_BANK_CONFLICT', 0) / wc:5.3f}"
    if i < len(fl) and i < len(wl):
        s += f" fetch={2 * fl[i].get('FETCH_SIZE', 0) / 1024:7.1f}MB(x2 corr) write={wl[i].get('WRITE_SIZE', 0) / 1024:7.1f}MB"
    print(s)
    i += 1
