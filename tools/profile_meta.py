"""Stamp a committed profile with the identity of the code it was taken with: writes <profile>.meta.json = {kernel_src_sha16 (sha256
over the named kernel sources of THIS tree, as bench.py computes it), git_head, date}.  bench.py reports a number read from a committed
profile only when the stamp matches the tree it runs from (a stale profile is dropped with the reason on the bench line).
Run in the build container right after copying the file from gpurun_out/ to profiles/, with the tree the gpurun call shipped.
usage: python tools/profile_meta.py profiles/<file> <csrc file> [<csrc file> ...]"""
import datetime, hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path, files = sys.argv[1], sys.argv[2:]
h = hashlib.sha256()
for f in files:
    h.update(open(os.path.join(ROOT, "ming_univision_amd", "csrc", f), "rb").read())
head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--"] + ["ming_univision_amd/csrc/" + f for f in files],
                            capture_output=True, text=True).stdout.strip())      # of the NAMED sources only
meta = {"kernel_src_sha16": h.hexdigest()[:16], "kernel_sources": files, "git_head": head + ("+uncommitted changes in these sources" if dirty else ""),
        "stamped": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ")}
json.dump(meta, open(path + ".meta.json", "w"), indent=1)
print(path + ".meta.json", meta)
