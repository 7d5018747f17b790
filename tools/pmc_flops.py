"""Executed floating-point operations per evaluation from rocprofv3 instruction counters (tools/pmc_run.sh summaries):
flops = (2 FMA + ADD + MUL + TRANS) x 64 lanes / batch.  Writes profiles/<round>_pmc_flops.json (read by bench.py).
usage: python tools/pmc_flops.py OUT.json  workload:algo:dtype:batch:summary.txt ..."""
import json, re, sys
out, entries = sys.argv[1], []
for spec in sys.argv[2:]:
    w, a, d, b, path = spec.split(":")
    c = {m.group(1): float(m.group(2)) for m in re.finditer(r"^(\S+)\s+n=\s*\d+\s+mean=(\S+)", open(path).read(), re.M)}
    sfx = "F32" if d == "f32" else "F64"
    fma, add, mul = c.get(f"SQ_INSTS_VALU_FMA_{sfx}", 0), c.get(f"SQ_INSTS_VALU_ADD_{sfx}", 0), c.get(f"SQ_INSTS_VALU_MUL_{sfx}", 0)
    trans = c.get("SQ_INSTS_VALU_TRANS_F32", 0) if d == "f32" else 0
    flops = (2 * fma + add + mul + trans) * 64 / int(b)
    entries.append({"workload": w, "algo": a, "dtype": d, "batch": int(b), "flops_per_eval": flops,
                    "fma": fma, "add": add, "mul": mul, "trans": trans, "valu": c.get("SQ_INSTS_VALU"),
                    "source": f"(2 FMA + ADD + MUL + TRANS) x 64 / batch, SQ_INSTS_VALU_* of {path.split('gpurun_out/')[-1]}"})
json.dump({"_comment": "executed flops per evaluation from rocprofv3 PMC (wave-instruction counts x 64 lanes; masked lanes of the "
                       "ragged last tile count as executed)", "entries": entries}, open(out, "w"), indent=1)
print(json.dumps(entries, indent=1))
