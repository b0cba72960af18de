"""Run in the build container (needs /root/reference): copies the reference's headline model files (BPRMF / NeuMF / SASRec and the
FM / WideDeep / DeepFM context family) VERBATIM into
tests/golden/reference_models/ -- category (b) test fixtures: the GPU box has no /root/reference, and the drop-in test
(tests/test_gpu_reference_heads.py) must feed the plugin the reference's OWN, unmodified files.  Nothing under rechorus_amd/
imports or reads them.  Also writes MANIFEST.json (sha256 of each file + the syntax-tree hash of its forward functions, which
rechorus_amd/dropin.py::KNOWN_FORWARD_HASHES must list) and checks that dropin.py lists them."""
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference/src/models"
OUT = os.path.join(ROOT, "tests", "golden", "reference_models")
FILES = (("general", "BPRMF", ("",)), ("general", "NeuMF", ("",)), ("sequential", "SASRec", ("",)),
         ("context", "FM", ("CTR", "TopK")), ("context", "WideDeep", ("CTR", "TopK")), ("context", "DeepFM", ("CTR", "TopK")))


def main():
    from rechorus_amd import dropin
    sys.path.insert(0, dropin.PLUGIN)
    import main as plugin_main
    manifest = {}
    for sub, name, modes in FILES:
        src = os.path.join(REF, sub, name + ".py")
        dst_dir = os.path.join(OUT, sub)
        os.makedirs(dst_dir, exist_ok=True)
        shutil.copyfile(src, os.path.join(dst_dir, name + ".py"))
        os.environ["RECHORUS_MODEL_DIRS"] = dst_dir
        hashes = {}
        for mode in modes:
            cls = plugin_main.find_class("model", (name, mode))
            h = dropin.forward_hash(cls)
            hashes[name + mode] = h
            print(name + mode, h, "listed" if h in dropin.KNOWN_FORWARD_HASHES[name + mode] else "NOT LISTED in rechorus_amd/dropin.py")
        entry = {"sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(),
                 "source": "THUwangcy/ReChorus src/models/%s/%s.py (verbatim)" % (sub, name)}
        if modes == ("",):
            entry["forward_hash"] = hashes[name]
        else:
            entry["forward_hashes"] = hashes
        manifest[sub + "/" + name + ".py"] = entry
    json.dump(manifest, open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
