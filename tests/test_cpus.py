"""chromap_amd/cpus.py: the processors a process may really use (affinity mask cut to the control group's CPU quota)"""
import os

from chromap_amd import cpus


def test_budget_is_positive_and_within_the_affinity_mask():
    n = cpus.cpu_budget()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def test_budget_override(monkeypatch):
    monkeypatch.setenv("CM_CPU_BUDGET", "5")
    assert cpus.cpu_budget() == 5
    monkeypatch.setenv("CM_CPU_BUDGET", "junk")
    assert cpus.cpu_budget() >= 1


def test_quota_cuts_the_budget(monkeypatch, tmp_path):
    """cgroup v2's cpu.max "<quota> <period>": 1600000 100000 = 16 CPUs (what the measurement boxes give a job that sees 256)"""
    import builtins
    real_open = builtins.open
    fake = tmp_path / "cpu.max"
    fake.write_text("200000 100000\n")

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            return real_open(fake, *a, **k)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.delenv("CM_CPU_BUDGET", raising=False)
    n_aff = len(os.sched_getaffinity(0))
    assert cpus.cpu_budget() == min(2, n_aff)
    fake.write_text("max 100000\n")
    assert cpus.cpu_budget() == n_aff
