"""Drop-in check: the REFERENCE's own test scripts and its MonotonicMLP.py (BASELINE config C0) run unchanged against this
package through the `compat/models` import shim.  Needs the read-only reference tree, which exists only in the build
container -- skipped elsewhere (nothing here is needed on the GPU box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tests")), reason="reference tree not mounted")


def _run(script, *args, timeout=900, tmp_path=None):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "compat"), ROOT]), MPLBACKEND="Agg",
               CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    return subprocess.run([sys.executable, os.path.join(REF, script), *args], cwd=str(tmp_path), env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_reference_test_jit_passes_against_this_package(tmp_path):
    r = _run("tests/test_jit.py", tmp_path=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "All tests completed successfully" in r.stdout
    assert "✗" not in r.stdout


def test_reference_numerical_validation_passes_against_this_package(tmp_path):
    r = _run("tests/test_numerical_validation.py", tmp_path=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "All numerical validation tests passed" in r.stdout
    for name in ("integral_convergence", "gradient_correctness", "monotonic_fitting", "integral_accuracy"):
        assert f"PASSED: {name}" in r.stdout


def test_reference_monotonic_mlp_script_runs_unchanged(tmp_path):
    r = _run("MonotonicMLP.py", "-nb_train", "300", "-nb_test", "100", "-nb_epoch", "1", tmp_path=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]


def test_reference_toy_experiment_runs_and_checkpoint_loads_into_the_reference(tmp_path):
    """ToyExperiments.py (BASELINE config C1's script) has no epoch flag: let it train for a while, kill it, and check
    it logged, plotted (which exercises UMNNMAFFlow.invert) and checkpointed.  The checkpoint written by THIS package
    must then load, strictly, into the REFERENCE's UMNNMAFFlow and give the same log-likelihood."""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "compat"), ROOT, REF]), MPLBACKEND="Agg",
               CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    try:
        subprocess.run([sys.executable, os.path.join(REF, "ToyExperiments.py"), "-dataset", "moons"], cwd=str(tmp_path),
                       env=env, capture_output=True, text=True, timeout=75)
    except subprocess.TimeoutExpired:
        pass
    out = tmp_path / "moons"
    assert (out / "model.pt").exists() and (out / "0.png").exists() and (out / "logs").exists()
    assert "epoch: 0" in (out / "logs").read_text()
    check = (
        "import sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "from models.UMNN import UMNNMAFFlow as RefFlow\n"
        "sys.path.insert(0, %r)\n"
        "import umnn_amd\n"
        "sd = torch.load(%r)\n"
        "ref = RefFlow(nb_flow=1, nb_in=2, hidden_derivative=[100]*4, hidden_embedding=[100]*4, embedding_s=10, nb_steps=20)\n"
        "ref.load_state_dict(sd)\n"
        "mine = umnn_amd.UMNNMAFFlow(nb_flow=1, nb_in=2, hidden_derivative=[100]*4, hidden_embedding=[100]*4, embedding_s=10, nb_steps=20)\n"
        "mine.load_state_dict(sd)\n"
        "torch.manual_seed(0); x = torch.randn(64, 2)\n"
        "with torch.no_grad():\n"
        "    a, _ = ref.compute_ll(x); b, _ = mine.compute_ll(x)\n"
        "assert torch.allclose(a, b, atol=1e-5, rtol=1e-5), (a - b).abs().max()\n"
        "print('checkpoint ok')\n") % (REF, ROOT, str(out / "model.pt"))
    r = subprocess.run([sys.executable, "-W", "ignore", "-c", check], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
    assert r.returncode == 0 and "checkpoint ok" in r.stdout, r.stderr[-2000:]


_VAE_SNIPPET = r"""
import sys, argparse, torch
torch.manual_seed(0)
from models.vae_lib.models import VAE
import models
args = argparse.Namespace(z_size=8, input_size=[1, 28, 28], input_type='binary', cuda=False, made_h_size=16, num_flows=2,
                          gpu_num=0, steps=20, solver=%(solver)r, hidden_embedding=[32, 32],
                          hidden_derivative=[50, 50, 50, 50], embedding_size=6)
vae = VAE.MMAVAE(args)
path = %(path)r
if %(load)r:
    vae.load_state_dict(torch.load(path + '/vae.pt'))
else:
    torch.save(vae.state_dict(), path + '/vae.pt')
torch.manual_seed(1)
x = torch.rand(5, 1, 28, 28).bernoulli()
x_mean, z_mu, z_var, ldj, z0, zk = vae(x)
loss = (x_mean - x).pow(2).sum() - ldj.sum() + zk.pow(2).sum()
loss.backward()
if %(load)r:
    vae.flow.model.force_lipschitz(1.5)      # the reference never defined the name its VAE wrapper calls (flows.py:327)
else:
    vae.forceLipshitz(1.5)                   # the shim provides it
out = dict(module=models.UMNNMAFFlow.__module__, x_mean=x_mean.detach(), ldj=ldj.detach(), zk=zk.detach(),
           grads={n: p.grad.clone() for n, p in vae.named_parameters() if p.grad is not None},
           after={n: p.detach().clone() for n, p in vae.flow.named_parameters()})
torch.save(out, path + '/%(tag)s.pt')
print('vae ok')
"""


@pytest.mark.parametrize("solver", ["CC", "CCParallel"])
def test_reference_vae_flow_prior_uses_this_package_and_matches_the_reference(tmp_path, solver):
    """BASELINE config C4's caller: the reference's MMAVAE (models/vae_lib, TrainVaeFlow.py) builds its conditional
    UMNNMAFFlow through `from models import UMNNMAFFlow` (flows.py:16,309) and calls compute_log_jac_bis, forceLipshitz
    (flows.py:318-330).  Through the shim it must construct, train one step and match the pure reference run on the
    same weights and noise (outputs and every parameter gradient)."""
    import torch
    def run(tag, pythonpath, load):
        code = _VAE_SNIPPET % dict(solver=solver, path=str(tmp_path), load=load, tag=tag)
        env = dict(os.environ, PYTHONPATH=os.pathsep.join(pythonpath), CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
        r = subprocess.run([sys.executable, "-W", "ignore", "-c", code], cwd=str(tmp_path), env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "vae ok" in r.stdout, r.stderr[-3000:]
        return torch.load(str(tmp_path / (tag + ".pt")))
    mine = run("mine", [os.path.join(ROOT, "compat"), ROOT, REF], False)
    ref = run("ref", [REF], True)
    assert mine["module"].startswith("umnn_amd") and not ref["module"].startswith("umnn_amd")
    for k in ("x_mean", "ldj", "zk"):
        assert torch.allclose(mine[k], ref[k], rtol=1e-4, atol=1e-5), (k, (mine[k] - ref[k]).abs().max())
    assert set(mine["grads"]) == set(ref["grads"])
    for n, g in ref["grads"].items():
        scale = g.abs().max().clamp_min(1e-6)
        assert (mine["grads"][n] - g).abs().max() <= 2e-4 * scale, n
    for n, p in ref["after"].items():
        assert torch.allclose(mine["after"][n], p, rtol=1e-5, atol=1e-6), n
