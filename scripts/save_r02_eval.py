"""Copy the summaries collected by scripts/collect_r02_eval.sh (gpurun_out/r02e) into profiles/."""
import glob, os, shutil
src, dst = "gpurun_out/r02e", "profiles"
shutil.copy(f"{src}/bench_other_workloads.jsonl", f"{dst}/r02_bench_other_workloads.jsonl")
stats = sorted(glob.glob(f"{src}/prof_eval/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1]
keep = [l for i, l in enumerate(open(stats)) if i == 0 or "yr::" in l]
open(f"{dst}/r02_eval_kernel_stats.csv", "w").writelines(keep)
shutil.copy(f"{src}/eval_split.txt", f"{dst}/r02_eval_precisions_k_prescan.txt")
shutil.copy(f"{src}/cdae_valid_lists.txt", f"{dst}/r02_cdae_validate_evaluate_ms.txt")
lines = [l for l in open(f"{src}/eval_phases.txt") if l.startswith("k=") or l.startswith("waves")]
open(f"{dst}/r02_eval_phase_cycles.txt", "w").writelines(lines)
print(open(f"{dst}/r02_eval_kernel_stats.csv").read()[:3000])
print(open(f"{dst}/r02_eval_phase_cycles.txt").read())
print(open(f"{dst}/r02_cdae_validate_evaluate_ms.txt").read())
