"""profiles/r09_tolerances.txt from the records a test run left with RC_TOL_REPORT=<file> (tests/conftest.py): per test function and
tolerance setting above the north star's 1e-5, the largest error actually observed and how much of the allowance it used.
    RC_TOL_REPORT=gpurun_out/<tag>/tol.jsonl python -m pytest tests -m gpu -q ; python tools/tolerance_report.py gpurun_out/<tag>/tol.jsonl"""
import collections
import json
import re
import sys


def main(path):
    groups = collections.OrderedDict()
    n = 0
    for line in open(path):
        r = json.loads(line)
        n += 1
        if max(r["rtol"], r["atol_scale"]) <= 1e-5 and r["abs_floor"] == 0:
            continue      # at the north star's tolerance: nothing to justify
        test = re.sub(r"\[.*", "", r["test"])
        what = re.sub(r"[0-9]+", "#", r["what"])[:48]
        key = (test, r["kind"], r["rtol"], r["atol_scale"], what)
        g = groups.setdefault(key, {"calls": 0, "used": 0.0, "rel": 0.0, "floor": 0.0})
        g["calls"] += 1
        g["used"] = max(g["used"], r["tolerance_used"])
        g["rel"] = max(g["rel"], r["max_err_over_scale"])
        g["floor"] = max(g["floor"], r["abs_floor"])
    print("# %d comparisons recorded; %d groups allow more than rtol = atol_scale = 1e-5 (or carry an absolute floor)" % (n, len(groups)))
    print("# used = largest observed |got - want| / allowed over all elements and calls of the group (1.0 = at the limit);")
    print("# max err / scale = largest observed |got - want| / max |want|.  Sorted by headroom (least first).")
    head = "%-92s %-6s %-8s %-8s %-9s %6s %10s %8s" % ("test :: what", "kind", "rtol", "atol", "floor", "calls", "err/scale", "used")
    for kind, title in (("close", "assert_close (values, gradients): rtol / atol as applied, i.e. after conftest.TOL_CAP"),
                        ("update", "assert_update_close (optimizer updates W - W0): `used` is computed over ALL elements -- the elements a test names "
                                   "as ill-conditioned under Adam (|g| < 1e-7, excluded) are zeroed, the <= 0.5 % outlier allowance of the non-strict "
                                   "mode is NOT applied here, so a value above 1 marks such outliers, not a failed comparison")):
        print("\n## " + title)
        print(head)
        for key, g in sorted(((k, v) for k, v in groups.items() if k[1] == kind), key=lambda kv: -kv[1]["used"]):
            test, _, rtol, atol, what = key
            print("%-92s %-6s %-8.0e %-8.0e %-9.1e %6d %10.2e %8.3f" % ((test.split("/")[-1] + " :: " + what)[:92], kind, rtol, atol, g["floor"],
                                                                        g["calls"], g["rel"], g["used"]))


if __name__ == "__main__":
    main(sys.argv[1])
