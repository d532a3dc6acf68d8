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
    # ONE option table: the engine's own (TWK_HIP_OPTIONS -> twk_hip_option_describe), printed into INTEGRATION.md between its markers
    import tomahawk_amd.hip as H
    table = H.option_table()
    assert len(table) >= 25 and len({k for k, *_ in table}) == len(table)
    for key, dflt, lo, hi, doc in table:
        assert lo <= dflt <= hi and len(doc) > 10, key
    guide = _read("INTEGRATION.md")
    block = guide[guide.index("<!-- engine options:"):guide.index("<!-- /engine options -->")]
    assert block.split("-->", 1)[1].strip() == H.option_table_markdown().strip(), "INTEGRATION.md's option table is stale: regenerate it with tomahawk_amd.hip.option_table_markdown()"
    # the header names every family of keys and points at the table; nothing else in the repository carries a second table of them
    header = _read("include", "twk_hip.h")
    assert "twk_hip_option_describe" in header and "INTEGRATION.md" in header
    assert not re.search(r'^ \*   "[a-z_]+"\s+-?\d+\s', header, re.M), "include/twk_hip.h carries an option table of its own again"


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
