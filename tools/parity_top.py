import json
r=json.load(open("gpurun_out/parity_report.json"))
for k,v in r["by_dtype_and_kind"].items(): print(k, v["n"], "max_rel", "%.2e"%v["max_rel_err"], "beyond", "%.2e"%v["max_beyond_final_rounding"], v["worst_beyond_final_rounding"])
top=sorted([a for a in r["all"] if a["dtype"]=="bfloat16"], key=lambda a:-a["beyond_final_rounding"])[:14]
for a in top: print("%.2e %.2e"%(a["beyond_final_rounding"],a["rel_err"]), a["test"][-70:], a["name"])
