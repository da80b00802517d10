"""Host logic of the GPU-resident PPO on CPU: the policy network against the reference's own checkpoint
tensors and recorded actions, the update arithmetic against a straightforward autograd restatement, the flat-gradient
exchange under gloo (world_size 2), checkpoint round trips, and the reference's best policy flying the CPU oracle env.
The HIP pieces (amenv_gae, amenv_gaussian_act, the rollout itself) are covered by tests/test_gpu_ppo.py."""
import copy
import glob
import math
import os
import socket
import sys
import zipfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import rl_aerial_manipulator_amd as amd
from rl_aerial_manipulator_amd.ppo import ActorCritic, ppo_update

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def fixture_state_dict():
    z = np.load(os.path.join(GOLD, "policy_2300000.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("_")}


def test_policy_has_the_reference_checkpoint_layout():
    """state_dict keys, order and shapes == policy.pth of the reference's checkpoints (30,537 parameters)."""
    sd = fixture_state_dict()
    pol = ActorCritic(20, 4)
    own = pol.state_dict()
    assert list(own.keys()) == list(sd.keys())
    assert all(tuple(own[k].shape) == tuple(sd[k].shape) for k in sd)
    assert pol.num_parameters() == 30537
    # SB3 orthogonal init: W W^T = gain^2 I on the smaller side, zero biases, log_std 0
    w = pol.mlp_extractor.policy_net[2].weight.detach()          # 64 x 128
    assert torch.allclose(w @ w.T, 2.0 * torch.eye(64), atol=1e-4)
    w = pol.action_net.weight.detach()
    assert torch.allclose(w @ w.T, 1e-4 * torch.eye(4), atol=1e-7)
    assert float(pol.log_std.abs().max()) == 0.0 and float(pol.value_net.bias.abs().max()) == 0.0


def test_reference_checkpoint_reproduces_recorded_actions():
    """The golden policy episodes hold the (clipped, deterministic) actions the reference checkpoint produced from the
    reference env's observations (tools/gen_golden.py): ActorCritic.from_sb3(...).predict must give the same actions."""
    pol = ActorCritic.from_sb3(fixture_state_dict())
    n = 0
    for f in sorted(glob.glob(os.path.join(GOLD, "policy_ep*.npz"))):
        g = np.load(f)
        obs_seen = np.concatenate([g["obs0"][None], g["obs"][:-1]])        # observation the action at step t was computed from
        a = pol.predict(torch.from_numpy(obs_seen)).numpy()
        assert np.abs(a - g["actions"]).max() < 2e-6, f
        n += len(a)
    assert n > 2000


def test_flat_buffers_alias_parameters_and_gradients():
    pol = ActorCritic(20, 4).flatten_()
    assert pol.flat_param.numel() == 30537
    obs = torch.randn(7, 20)
    pol.flat_grad.zero_()
    (pol.actor(obs).sum() + pol.critic(obs).sum() + pol.log_std.sum()).backward()
    g = torch.cat([p.grad.reshape(-1) for p in pol.parameters()])
    assert torch.equal(g, pol.flat_grad) and float(pol.flat_grad.abs().sum()) > 0
    before = pol.action_net.weight.detach().clone()
    pol.flat_param.data.add_(1.0)                                           # one tensor update moves every parameter
    assert torch.allclose(pol.action_net.weight.detach(), before + 1.0)


def _batch(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(n, 20, generator=g)
    actions = torch.randn(n, 4, generator=g) * 0.5
    old_logp = -3.0 + 0.1 * torch.randn(n, generator=g)
    adv = torch.randn(n, generator=g) * 3.0 + 1.0
    ret = torch.randn(n, generator=g) * 10.0
    return obs, actions, old_logp, adv, ret


def _plain_ppo_step(pol, opt, obs, actions, old_logp, adv, ret, clip=0.2, ent=5e-4, vf=0.5, max_norm=0.5, normalize=True):
    """SB3 PPO.train() for one minibatch, written the obvious way on an un-flattened copy of the policy."""
    if normalize:
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    dist = torch.distributions.Normal(pol.actor(obs), pol.log_std.exp())
    logp = dist.log_prob(actions).sum(-1)
    ratio = (logp - old_logp).exp()
    pl = -torch.min(adv * ratio, adv * ratio.clamp(1 - clip, 1 + clip)).mean()
    vl = ((ret - pol.critic(obs)) ** 2).mean()
    el = -dist.entropy().sum(-1).mean()
    opt.zero_grad()
    (pl + ent * el + vf * vl).backward()
    torch.nn.utils.clip_grad_norm_(pol.parameters(), max_norm)
    opt.step()


def test_update_matches_plain_autograd_restatement():
    torch.manual_seed(1)
    a = ActorCritic(20, 4)
    b = copy.deepcopy(a)
    a.flatten_()
    leaf = a.flat_param.requires_grad_(True)
    leaf.grad = a.flat_grad
    opt_a = torch.optim.Adam([leaf], lr=2e-4, eps=1e-5)
    opt_b = torch.optim.Adam(b.parameters(), lr=2e-4, eps=1e-5)
    data = _batch(512)
    for _ in range(3):   # three full-batch epochs == three plain steps
        _plain_ppo_step(b, opt_b, *data)
    rec = ppo_update(a, opt_a, *data, batch_size=512, n_epochs=3)
    for (k, va), vb in zip(a.state_dict().items(), b.state_dict().values()):
        assert torch.allclose(va, vb, rtol=1e-5, atol=1e-7), k
    assert all(math.isfinite(v) for v in rec.values()) and rec["grad_norm"] > 0


def test_minibatches_cover_the_buffer_once_per_epoch():
    """Shuffled minibatches: every sample is used exactly once per epoch (last batch may be short, as in SB3)."""
    seen = []

    class Spy(ActorCritic):
        def evaluate_actions(self, obs, actions):
            seen.append(obs[:, 0].detach().clone())
            return super().evaluate_actions(obs, actions)

    pol = Spy(20, 4).flatten_()
    leaf = pol.flat_param.requires_grad_(True)
    leaf.grad = pol.flat_grad
    obs, actions, old_logp, adv, ret = _batch(1000)
    obs[:, 0] = torch.arange(1000, dtype=torch.float32)
    ppo_update(pol, torch.optim.Adam([leaf], lr=1e-4), obs, actions, old_logp, adv, ret, batch_size=300, n_epochs=2)
    assert [len(s) for s in seen] == [300, 300, 300, 100] * 2
    for e in range(2):
        ids = torch.cat(seen[4 * e:4 * e + 4]).sort().values
        assert torch.equal(ids, torch.arange(1000, dtype=torch.float32))
    assert not torch.equal(seen[0], seen[4])


# ---- multi-rank gradient exchange (gloo, world_size 2) -------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import rl_aerial_manipulator_amd as amd_
    from rl_aerial_manipulator_amd.ppo import ActorCritic as AC, ppo_update as upd
    dist = amd_.sharding.init_process_group("gloo")
    torch.manual_seed(100 + rank)                      # deliberately different initial weights per rank ...
    pol = AC(20, 4).flatten_()
    dist.broadcast(pol.flat_param, src=0)              # ... made identical the way PPO.__init__ does it
    leaf = pol.flat_param.requires_grad_(True)
    leaf.grad = pol.flat_grad
    opt = torch.optim.SGD([leaf], lr=1e-3)             # SGD, no clipping: the parameters are linear in the exchanged gradient
    obs, actions, old_logp, adv, ret = _batch(512)     # (Adam's normalisation would hide a wrong scale)
    sl = slice(rank * 256, (rank + 1) * 256)           # each rank owns half of the samples
    upd(pol, opt, obs[sl], actions[sl], old_logp[sl], adv[sl], ret[sl], batch_size=256, n_epochs=2, normalize_advantage=False,
        max_grad_norm=None, dist=dist)
    torch.save(pol.flat_param.detach().clone(), os.path.join(out_dir, f"flat{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_update_equals_one_process_on_all_samples(tmp_path):
    """Average of the two shards' gradients == gradient over the union: after 2 epochs both ranks hold the parameters one
    process gets from the 512 samples (same initial weights = rank 0's)."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    f0, f1 = (torch.load(os.path.join(tmp_path, f"flat{r}.pt"), weights_only=True) for r in (0, 1))
    assert torch.equal(f0, f1)                          # ranks never diverge
    torch.manual_seed(100)
    pol = ActorCritic(20, 4).flatten_()
    leaf = pol.flat_param.requires_grad_(True)
    leaf.grad = pol.flat_grad
    p0 = pol.flat_param.detach().clone()
    ppo_update(pol, torch.optim.SGD([leaf], lr=1e-3), *_batch(512), batch_size=512, n_epochs=2, normalize_advantage=False, max_grad_norm=None)
    step, err = (pol.flat_param.detach() - p0), (pol.flat_param.detach() - f0)
    assert float(step.abs().max()) > 5e-4                                  # the update is far above the comparison tolerance
    assert float(err.abs().max()) < 1e-3 * float(step.abs().max())   # fp32 rounding of O(1) parameters is ~1e-7; a wrong scale would be ~1


# ---- checkpoints -------------------------------------------------------------------------------------------------------
def test_policy_pth_round_trip(tmp_path):
    pol = ActorCritic.from_sb3(fixture_state_dict())
    p = os.path.join(tmp_path, "policy.pth")
    pol.save_sb3_policy(p)
    sd = torch.load(p, weights_only=True)               # a plain dict of tensors with SB3's keys
    assert list(sd.keys()) == list(fixture_state_dict().keys())
    again = ActorCritic.from_sb3(p)
    obs = torch.randn(5, 20)
    assert torch.equal(again.predict(obs), pol.predict(obs))
    # an SB3-style archive (policy.pth inside a zip) is read the same way
    zp = os.path.join(tmp_path, "ppo_model.zip")
    with zipfile.ZipFile(zp, "w") as z:
        z.write(p, "policy.pth")
        z.writestr("data", "{}")
    assert torch.equal(ActorCritic.from_sb3(zp).predict(obs), pol.predict(obs))


def test_arm_policy_dimensions():
    pol = ActorCritic(29, 7)
    assert pol.action_low.tolist() == [0.0] + [-1.0] * 6 and pol.action_high.tolist() == [2.0] + [1.0] * 6
    a = pol.predict(torch.randn(3, 29) * 100)
    assert a.shape == (3, 7) and bool((a >= pol.action_low).all()) and bool((a <= pol.action_high).all())


# ---- the reference's trained policy on the restated environment -------------------------------------------------------
def test_reference_policy_flies_the_oracle_env():
    """Closed loop: the checkpoint the reference's README names as best (v2/README.md:55), trained on the reference env,
    reaches and holds the waypoint on the CPU restatement with fresh random resets -- observation semantics, action
    scaling, dynamics and the reach/hold/terminate state machine all have to agree for that.  (The same check runs on the
    HIP environment in tests/test_gpu_ppo.py.)"""
    from oracle import oracle as O
    pol = ActorCritic.from_sb3(fixture_state_dict())
    env = O.OracleEnv(O.reference_quad_config(64, seed=3))
    o = env.reset()
    eps = succ = 0
    rets, lens = [], []
    for _ in range(1700):
        out = env.step(pol.predict(torch.from_numpy(o)).numpy(), nthreads=4)
        o = out["obs"]
        d = out["done"] != 0
        if d.any():
            eps += int(d.sum())
            succ += int(((out["info"][d] & O.INFO_SUCCESS) != 0).sum())
            rets += list(out["ep_return"][d])
            lens += list(out["ep_len"][d])
    assert eps >= 100 and succ / eps > 0.95
    assert 12000 < np.mean(rets) < 26000                # README reports episode returns of this order for the checkpoint
    assert 500 < np.mean(lens) < 900                    # reach at 150-300 steps + 500-step hold


def test_tall_skinny_linear_gradients_match_nn_linear():
    """The slab-wise weight gradient (ragged last slab included) against torch's own Linear backward."""
    from rl_aerial_manipulator_amd.ppo import _Linear
    torch.manual_seed(3)
    lin = _Linear(20, 128).double()
    x = torch.randn(1000, 20, dtype=torch.float64, requires_grad=True)
    gy = torch.randn(1000, 128, dtype=torch.float64)
    y = lin(x)
    assert y.grad_fn is not None and "TallSkinny" in type(y.grad_fn).__name__
    y.backward(gy)
    x2 = x.detach().clone().requires_grad_(True)
    w = lin.weight.detach().clone().requires_grad_(True)
    b = lin.bias.detach().clone().requires_grad_(True)
    torch.nn.functional.linear(x2, w, b).backward(gy)
    assert torch.allclose(y, torch.nn.functional.linear(x.detach(), w.detach(), b.detach()), rtol=1e-12, atol=1e-12)
    assert torch.allclose(lin.weight.grad, w.grad, rtol=1e-11, atol=1e-11)
    assert torch.allclose(lin.bias.grad, b.grad, rtol=1e-11, atol=1e-11)
    assert torch.allclose(x.grad, x2.grad, rtol=1e-11, atol=1e-11)


def test_evaluate_policy_counts_episodes_like_sb3():
    """evaluate_policy on the oracle-backed stand-in env: every env contributes ceil(n_eval / num_envs) episodes; mean / std of
    their returns (population std, as np.std in SB3)."""
    from rl_aerial_manipulator_amd.ppo import evaluate_policy
    from tests.oracle_backend import OracleBackend
    pol = ActorCritic.from_sb3(fixture_state_dict())
    env = OracleBackend(8, seed=4)
    seen = []
    step = env.step

    def spy(a):
        out = step(a)
        d = out[2].numpy().astype(bool)
        seen.extend((i, float(env.ep_return[i])) for i in np.nonzero(d)[0])
        return out
    env.step = spy
    mean, std = evaluate_policy(pol, env, n_eval_episodes=10, check_every=16)     # -> 2 episodes per env
    first = {}
    for i, r in seen:
        first.setdefault(i, [])
        if len(first[i]) < 2:
            first[i].append(r)
    rs = np.array([r for v in first.values() for r in v])
    assert len(rs) == 16 and abs(mean - rs.mean()) < 1e-6 * abs(rs.mean()) and abs(std - rs.std()) < 1e-3 * max(1.0, rs.std())
    assert 10000 < mean < 26000


def test_three_way_bf16_split_is_exact_and_six_products_are_fp32_grade():
    """The arithmetic of csrc/amenv_mlp_train.hpp restated in numpy (split_pair_packed / the product list of prod_a, prod_b): x = hi + mid + lo
    holds EXACTLY with every part a bf16 value (8 significant bits, rounded to nearest even as v_cvt_pk_bf16_f32 does), whatever the magnitude
    or sign, and the six partial products of weight >= 2^-16 reproduce a b to 2^-23 relative -- an fp32 rounding, which is why the autograd
    gate did not move.  Truncated parts (same instruction count) would drop up to 2^-21; a two-part split 2^-15."""
    rng = np.random.RandomState(0)
    x = np.concatenate([rng.standard_normal(20000), rng.standard_normal(20000) * 1e-6, rng.standard_normal(20000) * 1e6,
                        np.float32([1.0, -1.0, 0.0, 3.0e38, 1.2e-30, 0.99999994, -2.0000002])]).astype(np.float32)

    def bf16(v, rne=True):
        u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
        if rne:
            u = u + 0x7FFF + ((u >> 16) & 1)
        return (u & 0xFFFF0000).astype(np.uint32).view(np.float32)

    def split3(v, rne=True):
        hi = bf16(v, rne)
        r1 = (v - hi).astype(np.float32)
        mid = bf16(r1, rne)
        lo = (r1 - mid).astype(np.float32)
        return hi, mid, lo

    def six(pa, pb):
        return sum(pa[i].astype(np.float64) * pb[j].astype(np.float64) for i, j in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)))

    a, b = x[:60000:3], x[1:60000:3]
    exact = a.astype(np.float64) * b.astype(np.float64)
    err = {}
    for rne in (True, False):
        hi, mid, lo = split3(x, rne)
        for part in (hi, mid, lo):
            assert np.all((part.view(np.uint32) & np.uint32(0xFFFF)) == 0)          # representable in bf16: the low 16 bits are zero
        assert np.array_equal((hi.astype(np.float64) + mid.astype(np.float64)) + lo.astype(np.float64), x.astype(np.float64))   # exact
        err[rne] = (np.abs(six(split3(a, rne), split3(b, rne)) - exact) / np.maximum(np.abs(exact), 1e-300)).max()
    assert err[True] < 2.0 ** -23 and 2.0 ** -23 < err[False] < 2.0 ** -21, err
    pa, pb = split3(a), split3(b)
    two_part = sum(pa[i].astype(np.float64) * pb[j].astype(np.float64) for i, j in ((0, 0), (0, 1), (1, 0)))                    # what a hi + mid split would give
    assert (np.abs(two_part - exact) / np.maximum(np.abs(exact), 1e-300)).max() > 2.0 ** -17
