"""Writes tests/golden/fp32_kernel_errors.json: today's max / rms error of every fp32 contraction kernel against torch's fp64 convolution
on the seeded cases of tests/fp32_error_cases.py (run on an MI355X; tests/test_gpu_fp32_errors.py asserts <= 4 x these).  --only-missing: measure only the cases the file does not hold yet."""
import importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fp32_error_cases as fc
L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib()
out = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.join(ROOT, "tests", "golden", "fp32_kernel_errors.json")
table = {"device": torch.cuda.get_device_name(0), "unit": "max|err|/max|ref|, rms(err)/rms(ref) against torch fp64", "cases": {}}
keep = "--only-missing" in sys.argv and os.path.exists(out)            # add new cases without re-basing (loosening) the existing budgets
if keep:
    table = json.load(open(out))
for fam, shape, seed in fc.CASES:
    if keep and fc.case_key(fam, shape, seed) in table["cases"]:
        continue
    mx, rms = fc.run_case(L, fam, shape, seed)
    table["cases"][fc.case_key(fam, shape, seed)] = {"max": mx, "rms": rms}
    print("%-44s max %.3e rms %.3e" % (fc.case_key(fam, shape, seed), mx, rms), flush=True)
json.dump(table, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out)
