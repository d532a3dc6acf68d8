"""Documentation that has to stay in step with the code (CPU only)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(*parts):
    with open(os.path.join(ROOT, *parts), encoding="utf-8", errors="replace") as fh:
        return fh.read()


def test_every_environment_switch_of_the_code_is_in_the_integration_guide():
    """INTEGRATION.md's table of environment variables names every TWK_* variable the sources read."""
    names = set()
    src_dirs = [os.path.join(ROOT, "tomahawk_amd", "csrc", d) for d in ("hip", "host")]
    files = [os.path.join(d, f) for d in src_dirs for f in os.listdir(d)]
    files += [os.path.join(ROOT, "bench.py")] + [os.path.join(ROOT, "tomahawk_amd", f) for f in os.listdir(os.path.join(ROOT, "tomahawk_amd")) if f.endswith(".py")]
    for path in files:
        text = _read(path)
        names |= set(re.findall(r'getenv\("(TWK_[A-Z0-9_]+)"\)', text))
        names |= set(re.findall(r'environ(?:\.get)?[\(\[]\s*"(TWK_[A-Z0-9_]+)"', text))
    assert len(names) >= 7
    guide = _read("INTEGRATION.md")
    missing = sorted(n for n in names if n not in guide)
    assert not missing, f"not in INTEGRATION.md: {missing}"


def test_the_engine_library_reads_no_environment_and_its_switches_are_documented():
    """libtwk_hip.so's sources contain no getenv at all (its switches are twk_hip_set_option keys); the host library reads
    only placement and the two behavioural switches; every option key is in include/twk_hip.h and INTEGRATION.md."""
    hip_dir = os.path.join(ROOT, "tomahawk_amd", "csrc", "hip")
    for f in os.listdir(hip_dir):
        assert "getenv" not in _read(hip_dir, f), f
    host_dir = os.path.join(ROOT, "tomahawk_amd", "csrc", "host")
    host_env = set()
    for f in os.listdir(host_dir):
        host_env |= set(re.findall(r'getenv\("([A-Z0-9_]+)"\)', _read(host_dir, f)))
    assert host_env == {"TWK_HIP_DEVICE", "TWK_HIP_GPUS", "TWK_HIP_PART", "TWK_REF_COMPAT", "TWK_HIP_NO_SCREEN"}, host_env
    keys = re.findall(r'\{"([a-z_]+)", &Options::', _read(hip_dir, "twk_hip.hip"))
    assert len(keys) >= 13
    header, guide = _read("include", "twk_hip.h"), _read("INTEGRATION.md")
    for k in keys:
        stem = k[:-5] if k.endswith(("_rows", "_cols")) else k          # "patch_rows/_cols" is one line of the header's table
        assert stem in header and k in guide, k


def test_profiles_readme_names_only_files_that_exist():
    """profiles/README.md: every `r0N_*` file it cites is in profiles/ (brace lists expanded by hand here)."""
    text = _read("profiles", "README.md")
    have = set(os.listdir(os.path.join(ROOT, "profiles")))
    cited = set(re.findall(r"`(r0\d_[A-Za-z0-9_.\-]+\.(?:json|txt|csv|log))`", text))
    assert len(cited) > 20
    missing = sorted(c for c in cited if c not in have)
    assert not missing, f"cited in profiles/README.md but absent: {missing}"


def test_design_and_readme_cite_only_profile_files_that_exist():
    have = set(os.listdir(os.path.join(ROOT, "profiles")))
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        cited = set(re.findall(r"`(?:profiles/)?(r0\d_[A-Za-z0-9_.\-]+\.(?:json|txt|csv|log))`", _read(doc)))
        missing = sorted(c for c in cited if c not in have)
        assert not missing, f"{doc} cites {missing}, not in profiles/"
