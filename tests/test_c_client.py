"""A host written in plain C against include/epi.h alone (tests/c_client): the
header compiles as strict C99, every symbol it uses links, and on the device it
produces the same numbers as the Python binding."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CDIR = os.path.join(ROOT, "tests", "c_client")
EXE = os.path.join(CDIR, "filter_client")


def test_header_is_strict_c99_and_client_links():
    subprocess.check_call(["make", "-C", CDIR, "-B", "filter_client"], stdout=subprocess.DEVNULL)
    out = subprocess.run(["ldd", EXE], capture_output=True, text=True, check=True).stdout
    line = [l for l in out.splitlines() if "libepi_gfx950.so" in l]
    assert line and "not found" not in line[0]
    # usage error path needs no device
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr


def _params_text(pomdp):
    c = pomdp.to_c()
    H = c.H
    vals = [c.H, c.D, c.N, c.inf_loss, c.test_loss, c.testrate_loss]
    for tab in (c.q, c.p, c.k, c.theta):
        vals += [float(tab[i]).hex() for i in range(H)]
    return " ".join(str(v) for v in vals) + "\n"


@pytest.mark.gpu
@pytest.mark.parametrize("delay", [0, 2])
def test_c_host_matches_python_binding(delay):
    import epidemicmodeling_amd as em
    assert os.path.exists(EXE), "build tests/c_client/filter_client before the GPU run (__graft_entry__.build does)"
    pomdp = em.make_pomdp(LOD=5, test_delay=delay)
    n, seed, a = 50_000, 4242, 0.25
    obs = [5100, 4900, 4700, 4400]
    r = subprocess.run([EXE, str(n), str(seed), repr(a)] + [str(o) for o in obs], input=_params_text(pomdp),
                       capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stderr
    lines = {l.split()[0] + (l.split()[1] if l.startswith("update") else ""): l.split() for l in r.stdout.splitlines()}
    assert lines["rows"][1] == str(2 + 14 + delay * 13) and lines["rows"][3] == str(n)

    b = em.initialbelief(pomdp, n, seed=seed)
    upd = em.BootstrapFilter(pomdp, n, seed=seed)
    for t, o in enumerate(obs):
        em.update(upd, b, em.Action(a), o)
        tok = lines[f"update{t}"]
        assert float.fromhex(tok[2]) == upd.last_loglik and float.fromhex(tok[3]) == upd.last_ess
        assert int(tok[4]) == upd.last_flags
    m = em.mean(b)
    got = [int(x) for x in lines["mean"][1:]]
    assert got[0] == m.S and got[1] == m.R and got[2:] == [int(x) for x in m.I]
    q = em.suminf_quantiles(b, [0.05, 0.5, 0.95])
    assert [int(x) for x in lines["quantiles"][1:]] == [int(x) for x in q]
    b.close()


@pytest.mark.gpu
def test_c_host_drives_the_sharded_update():
    """The same run as three shards behind epi_mctx, from plain C: diagnostics and belief mean equal the one-device
    run's (ESS to rounding: the sums of squares are added per shard)."""
    import epidemicmodeling_amd as em
    pomdp = em.make_pomdp(LOD=5, test_delay=1)
    n, seed, a = 50_001, 4242, 0.25
    obs = [5100, 4900, 9900]
    r = subprocess.run([EXE, "shards=3", str(n), str(seed), repr(a)] + [str(o) for o in obs], input=_params_text(pomdp),
                       capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stderr
    lines = {l.split()[0] + (l.split()[1] if l.startswith("update") else ""): l.split() for l in r.stdout.splitlines()}
    assert lines["rows"][3] == str(n)
    b = em.initialbelief(pomdp, n, seed=seed)
    upd = em.BootstrapFilter(pomdp, n, seed=seed)
    for t, o in enumerate(obs):
        em.update(upd, b, em.Action(a), o)
        tok = lines[f"update{t}"]
        assert float.fromhex(tok[2]) == upd.last_loglik and int(tok[4]) == upd.last_flags
        assert abs(float.fromhex(tok[3]) - upd.last_ess) <= 1e-9 * upd.last_ess
    m = em.mean(b)
    got = [int(x) for x in lines["mean"][1:]]
    assert got[0] == m.S and got[1] == m.R and got[2:] == [int(x) for x in m.I]
    b.close()
