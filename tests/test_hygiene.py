"""What travels to the GPU box: one-off probes and the records of rejected designs stay behind (round-2 review, item 7)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scratch_and_experiments_do_not_travel():
    lines = [l.strip() for l in open(os.path.join(ROOT, ".gpurunignore")) if l.strip() and not l.startswith("#")]
    assert "tools/scratch/" in lines and "tools/experiments/" in lines
    # ... and nothing that runs on the box reads them
    for rel in ("bench.py", "__graft_entry__.py", "tools/make_profiles.sh", "tools/report.py", "gpu-wah_amd/api.py"):
        text = open(os.path.join(ROOT, rel)).read()
        assert "tools/scratch" not in text and "tools/experiments" not in text, rel
    for name in os.listdir(os.path.join(ROOT, "tests")):
        if name.endswith(".py") and name != "test_hygiene.py":
            text = open(os.path.join(ROOT, "tests", name)).read()
            assert "tools/scratch" not in text and "tools/experiments" not in text, name
