"""Print the sha256 of the forward syntax trees of the reference's model files for the heads of the hot path -- the table
`rechorus_amd/dropin.py:KNOWN_FORWARD_HASHES` holds.  Run in the build container (needs /root/reference; nothing is copied: the files
are imported where they lie and only the hashes leave).
    python tests/golden/make_reference_forward_hashes.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/src/models"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rechorus_amd", "rechorus"))


def main():
    import main as plugin_main
    from rechorus_amd import dropin
    out = {}
    for sub, name, modes in (("general", "BPRMF", ("",)), ("general", "NeuMF", ("",)), ("sequential", "SASRec", ("",)),
                             ("context", "FM", ("CTR", "TopK")), ("context", "WideDeep", ("CTR", "TopK")), ("context", "DeepFM", ("CTR", "TopK"))):
        os.environ["RECHORUS_MODEL_DIRS"] = os.path.join(REF, sub)
        for mode in modes:
            cls = plugin_main.find_class("model", (name, mode))
            assert cls.forward.__globals__["__file__"].startswith(REF) or cls.__init__.__globals__["__file__"].startswith(REF)
            out[name + mode] = dropin.forward_hash(cls)
    print(json.dumps(out, indent=1))
    bad = {k: v for k, v in out.items() if v not in dropin.KNOWN_FORWARD_HASHES[k]}
    print("all listed in dropin.KNOWN_FORWARD_HASHES" if not bad else "NOT listed: %s" % bad)


if __name__ == "__main__":
    main()
