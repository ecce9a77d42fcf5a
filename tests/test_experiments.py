"""tools/experiments/*.patch hold kernel experiments and the wrong-result profiling probes that were taken OUT of the
product sources: they must keep applying to the current jetracer-orbslam2_amd/csrc, or the evidence
scripts that build from them (tools/build_variant.sh -p, tools/phase_counters.sh) rot silently.  No GPU, no compile."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCHES = sorted(glob.glob(os.path.join(ROOT, "tools", "experiments", "*.patch")))


@pytest.mark.parametrize("patch", PATCHES, ids=[os.path.basename(p) for p in PATCHES])
def test_experiment_patch_applies_to_the_product_sources(patch, tmp_path):
    dst = tmp_path / "jetracer-orbslam2_amd"
    dst.mkdir()
    shutil.copytree(os.path.join(ROOT, "jetracer-orbslam2_amd", "csrc"), dst / "csrc", ignore=shutil.ignore_patterns(".obj", ".pytest_cache"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    r = subprocess.run(["patch", "-p1", "--dry-run", "-i", patch], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_product_sources_carry_no_wrong_result_probe():
    """The macros whose builds produce deliberately wrong results (or other schedules) live only in the patches."""
    pat = re.compile(r"ORBFE_\w*(NOLDS|ROWS3|STOP_AFTER|ABLATE|NOSTAGE|SCRAMBLE|NO_WB)")
    hits = []
    for f in glob.glob(os.path.join(ROOT, "jetracer-orbslam2_amd", "csrc", "*")):
        if os.path.isfile(f) and not f.endswith((".o", ".so")):
            for i, line in enumerate(open(f, errors="replace"), 1):
                if pat.search(line):
                    hits.append("%s:%d" % (os.path.basename(f), i))
    assert not hits, hits
