"""BASELINE.json configs[3] (FSD) and configs[4] (FSDv2) as CHAINS ("a chain-level FSD / FSDv2 golden").

The wiring is bench_workloads.FSDPath / FSDv2Path - one piece of code over a module provider:
  * the reference's own modules (oracle/ref_fsd.py, build container) produced tests/golden/fsd_chain.npz and
    fsdv2_chain.npz: VoteSegmentor.extract_feat -> Voxel2PointScatterNeck -> ClusterAssigner -> SingleStageFSD.extract_feat
    (detectors/single_stage_fsd.py:228-250, 467-483, 922-999) and SingleStageFSDV2.extract_feat
    (single_stage_fsd_v2.py:159-271), forward + backward, training mode, fixture size;
  * the CPU port (oracle/fsd_cpu.py = bench.py's cpu_baseline for these workloads) is checked against the goldens on any box
    and against the live reference in the build container (other seed, eval mode);
  * the GPU path (sst_amd) is checked against the goldens (`-m gpu`): integer outputs - voxels, foreground selection,
    cluster assignment, virtual voxels - exactly, features 1e-3 (measured ~1e-5), gradients 1e-3 relative."""
import numpy as np
import pytest
import torch

from conftest import load_golden

import bench_workloads as BW
from oracle import ref_loader

CASES = {'fsd': (BW.FSDPath, BW.FSD_SMALL_CFG, dict(roi_stage=False), 18.0),
         'fsdv2': (BW.FSDv2Path, BW.FSDV2_SMALL_CFG, dict(), 12.0)}
INT_KEYS = ('voxel_coors', 'sel', 'cluster_inds', 'cluster_coors', 'virtual_coors')
# The NAMED decision-sensitive parameters of the GPU chains on the golden fixtures (a list, not a blanket
# allowance).  Everything else is held to 1e-3 of its gradient's scale or 4 x the reference's own fp32 noise on that parameter
# (measured on the FSD fixture: every parameter <= 2e-5; gpurun_out/chain_grad_errs_*.json).
#   fsdv2 / virtual_stage.recover_proj.0.0.weight: 1.4e-3 measured.  profiles/r04/chain_grad_isolation.txt: the kernels reproduce
#   this module's gradient to 1e-6 GIVEN its inputs (evaluated in float64 on the GPU's own input and upstream gradient); its
#   input differs by 2-3e-5 from the CPU formulation (the output-stationary convolution sums 27 x C_in products in one fp32
#   chain), which flips the sign of a handful of ReLU inputs next to zero; each flip moves an entry by |dy| |x|.
DECISION_SENSITIVE = {('fsdv2', 'virtual_stage.recover_proj.0.0.weight'): 5e-3}


def _build(tag, ops, golden, dev='cpu'):
    cls, cfg, kw, _ = CASES[tag]
    net = cls(ops, cfg, **kw)
    net.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in golden.items() if k.startswith('w::')}, strict=True)
    return net.to(dev).train()


def _check_against_golden(tag, net, g, dev, feat_tol, grad_tol):
    clouds = [torch.from_numpy(g['in::points0']).to(dev), torch.from_numpy(g['in::points1']).to(dev)]
    loss, stats, out = net(clouds, return_tensors=True)
    loss.backward()
    for k, v in stats.items():
        assert int(v) == int(g['stats::' + k]), (k, int(v), int(g['stats::' + k]))
    for key in [k[5:] for k in g if k.startswith('out::') and k != 'out::loss']:
        got = out[key][::4] if key == 'head' else out[key]
        want = g['out::' + key]
        if key in INT_KEYS:
            np.testing.assert_array_equal(got.cpu().numpy().astype(np.int64), want.astype(np.int64), err_msg=key)
        else:
            err = float(np.abs(got.detach().cpu().numpy() - want).max())
            assert err <= feat_tol * max(1.0, float(np.abs(want).max())), (key, err)
    assert abs(float(loss.detach()) - float(g['out::loss'])) <= feat_tol * max(1.0, abs(float(g['out::loss'])))
    params = dict(net.named_parameters())
    grad_keys = [k[6:] for k in g if k.startswith('grad::')]
    assert len(grad_keys) >= 8
    report = {}
    for key in grad_keys:
        # against the float64 evaluation stored beside the reference's fp32 gradient; the bar is the tolerance or the
        # reference's own fp32 deviation from float64 on that parameter, whichever is larger
        ref32, exact = g['grad::' + key].astype(np.float64), g['grad64::' + key]
        scale = max(1.0, float(np.abs(exact).max()))
        err = float(np.abs(params[key].grad.cpu().numpy().astype(np.float64) - exact).max())
        noise_ref = float(np.abs(ref32 - exact).max())
        report[key] = (err / scale, noise_ref / scale)
        # per parameter: a parameter the reference's own fp32 backward holds to 2.5e-4 of float64 is not
        # decision-sensitive on this fixture - it keeps the 1e-3 bar whatever `grad_tol` says; the looser floor is for the
        # parameters that are already noisy in the reference's fp32 evaluation
        floor = min(grad_tol, 1e-3) if noise_ref <= 2.5e-4 * scale else grad_tol
        floor = max(floor, DECISION_SENSITIVE.get((tag, key), 0.0) if dev != 'cpu' else 0.0)
        assert err <= max(floor * scale, 4.0 * noise_ref), (key, err / scale, noise_ref / scale)
    import json
    import os
    log_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if os.path.isdir(log_dir) and dev != 'cpu':
        with open(os.path.join(log_dir, f'chain_grad_errs_{tag}.json'), 'w') as f:
            json.dump({k: {'err_over_scale': a, 'reference_fp32_noise_over_scale': b} for k, (a, b) in report.items()}, f, indent=1)


@pytest.mark.parametrize('tag', ['fsd', 'fsdv2'])
def test_cpu_port_chain_matches_reference_golden(tag):
    from oracle import fsd_cpu
    g = load_golden(f'{tag}_chain.npz')
    _check_against_golden(tag, _build(tag, fsd_cpu, g), g, 'cpu', 1e-4, 1e-4)


@pytest.mark.skipif(not ref_loader.available(), reason='needs the reference tree (build container)')
@pytest.mark.parametrize('tag', ['fsd', 'fsdv2'])
@pytest.mark.parametrize('train', [True, False])
def test_cpu_port_chain_matches_live_reference(tag, train):
    """other weights, other clouds, training and eval mode (running batch-norm statistics, single-batch clustering),
    every gradient of the chain"""
    from oracle import fsd_cpu, ref_fsd
    cls, cfg, kw, half = CASES[tag]
    torch.manual_seed(7 + int(train))
    ref = cls(ref_fsd.reference_ops(), cfg, **kw)
    gen = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=gen) * 0.1)
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.2, generator=gen)
                m.running_var.uniform_(0.5, 1.5, generator=gen)
    port = cls(fsd_cpu, cfg, **kw)
    port.load_state_dict(ref.state_dict(), strict=True)
    ref.train(train)
    port.train(train)
    clouds = [BW.chain_cloud(3500, 11, half_extent=half)] if not train else \
        [BW.chain_cloud(3000, 12, half_extent=half), BW.chain_cloud(2500, 13, half_extent=half)]
    lr, sr, tr = ref(clouds, return_tensors=True)
    lp, sp, tp = port(clouds, return_tensors=True)
    assert sr == sp
    for key, a in tr.items():
        if a is None:
            continue
        if a.is_floating_point():
            assert float((a - tp[key]).abs().max()) <= 1e-4 * max(1.0, float(a.abs().max())), key
        else:
            assert torch.equal(a.long(), tp[key].long()), key
    if train:
        lr.backward()
        lp.backward()
        # two fp32 evaluations of one function agree to the fp32 evaluation noise of that function, which is NOT 1e-4 for
        # every parameter of these chains: batch norm over a few thousand rows followed by ReLU / max pooling makes single
        # parameters' gradients sensitive to a handful of decisions (tools/fsd_grad_adjudicate.py).  The noise is measured:
        # the same port evaluated in float64 (integer stages in fp32, so the decisions upstream are the same).
        port64 = cls(fsd_cpu, cfg, **kw).double()
        port64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in ref.state_dict().items()}, strict=True)
        port64.train(train)
        port64([c.double() for c in clouds])[0].backward()
        theirs, ours, exact = dict(ref.named_parameters()), dict(port.named_parameters()), dict(port64.named_parameters())
        checked, report = 0, []
        for name, p in theirs.items():
            if p.grad is None:
                assert ours[name].grad is None, name
                continue
            scale = max(1.0, float(p.grad.abs().max()))
            noise_port = float((ours[name].grad.double() - exact[name].grad).abs().max())     # fp32 port vs the exact gradient
            noise_ref = float((p.grad.double() - exact[name].grad).abs().max())               # fp32 REFERENCE vs the exact gradient
            err = float((p.grad - ours[name].grad).abs().max())
            # PER PARAMETER: forward outputs agree to 1e-6 (above); two correct fp32 evaluations of one gradient
            # differ by at most the sum of their distances from the exact one, and the float64 run measures both distances for
            # THIS parameter.  The reference is the noisier of the two (naiveSyncBN differentiates var = E[x^2] - mean^2 in fp32:
            # up to 7e-3 from float64 on the batch-norm-fed parameters, the port's F.batch_norm 1e-3); a defect of the port - a
            # mis-routed bias or norm gradient - is O(scale) on a parameter whose measured noise is 1e-6..1e-4 and fails here,
            # where the blanket 2e-2 / "15 % may miss" of round 4 let it pass.  Floor 2e-4 of the gradient's scale.
            bar = max(2e-4 * scale, 2.0 * (noise_port + noise_ref))
            report.append((name, err / scale, noise_port / scale, noise_ref / scale))
            assert err <= bar, (name, err / scale, noise_port / scale, noise_ref / scale)
            assert err <= 2e-2 * scale, (name, err / scale)        # and never the O(scale) of a wrong formula, whatever the noise
            checked += 1
        assert checked > 40
        # the well-conditioned half of the parameters is held to the tight bar by construction: say how many there are
        tight = sum(1 for _, e, a, b in report if 2.0 * (a + b) <= 2e-4)
        assert tight >= 0.2 * checked, (tight, checked)       # measured: 34 of 149 (FSDv2 chain), more on FSD's


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['fsd', 'fsdv2'])
def test_gpu_chain_matches_reference_golden(tag):
    g = load_golden(f'{tag}_chain.npz')
    # features 1e-3 (measured <= 3e-5).  Gradients 5e-3: the kernels reproduce every module's gradient to ~1e-6 GIVEN the module's
    # inputs (profiles/r04/chain_grad_isolation.txt: recover_proj evaluated in float64 on the GPU's own input and upstream
    # gradient), but a forward difference of 2-3e-5 - the output-stationary convolution sums 27 x C_in products in one fp32
    # chain, the CPU formulation per offset - flips the sign of a handful of ReLU inputs next to zero, and each flip moves a
    # gradient entry by |dy| |x|: 1e-3-class differences on single parameters, in either direction (the reference's own fp32
    # gradients sit up to 3e-3 from float64 on this fixture, stored beside them as grad64).  The statistical statement over
    # ALL parameters at 40 000 points is test_gpu_gradients_within_fp32_noise_at_40k.
    _check_against_golden(tag, _build(tag, BW.GpuOps, g, 'cuda:0'), g, 'cuda:0', 1e-3, 5e-3)


@pytest.mark.gpu
def test_gpu_gradients_within_fp32_noise_at_40k():
    """the FULL-WIDTH FSD chain (bench_workloads.FSD_CFG: 34 sparse convolutions, 3 + 2 SIR blocks, RoI
    stage) on a 40 000-point frame - every parameter gradient of the GPU path against the float64 evaluation of the CPU port
    (tests/adjudicate_fsd_grads.py), beside the fp32 evaluation of the same port.  The chain's gradients are ill-conditioned
    (training-mode batch norm over few rows, ReLU / max decisions: one fp32 rounding step on the weights moves single
    parameters' exact gradient by 1e-4..4e-3), so the bar is the fp32 noise of the reference algorithm itself:
      * integer stages (voxels, foreground, clusters, pooled pairs) equal;
      * per parameter: GPU within max(1e-3, 5 x the fp32 port's own distance) of float64, for >= 95 % of the parameters
        (each distance is ONE realisation of rounding noise, not a bound);
      * over all parameters the GPU is not further from float64 than the fp32 port is (median ratio <= 2, worst <= 3 x worst)."""
    import adjudicate_fsd_grads as ADJ
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, threads))
    try:
        net, cloud = ADJ.build('fsd', BW.GpuOps, 40000)
        net = net.to('cuda:0')
        loss, stats = net([cloud.to('cuda:0')])
        loss.backward()
        torch.cuda.synchronize()
        grads = {n: ADJ._sub(p.grad).cpu() for n, p in net.named_parameters() if p.grad is not None}
        res = ADJ.adjudicate('fsd', 40000, grads, {k: int(v) for k, v in stats.items()}, log=lambda *_: None, sensitivity=False)
    finally:
        torch.set_num_threads(threads)
    assert res['integer_stages_equal_gpu_port32'] and res['integer_stages_equal_port32_port64'], res['sizes']
    rows = res['rows']
    assert len(rows) > 150
    out = [(n, r) for n, r in rows.items() if r['gpu_vs_f64'] > max(1e-3, 5.0 * r['port32_vs_f64'])]
    assert len(out) <= 0.05 * len(rows), out
    assert res['gpu_error_over_port32_error']['median'] <= 2.0, res['gpu_error_over_port32_error']
    assert res['max_over_parameters']['gpu_vs_f64'] <= 3.0 * res['max_over_parameters']['port32_vs_f64'], res['max_over_parameters']
