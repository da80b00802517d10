"""GPU-resident PPO (BASELINE config 5) on the MI355X: the two HIP kernels of the loop against their
restatements, the rollout buffer against the environment and the policy, the reference's best checkpoint flying the HIP
environment, and a short training run."""
import math
import os

import numpy as np
import pytest
import torch

import rl_aerial_manipulator_amd as amd
from rl_aerial_manipulator_amd import _lib as L
from rl_aerial_manipulator_amd.ppo import ActorCritic, PPO, RolloutBuffer, compute_gae, gaussian_act

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def gae_magnitude(r, v, d, lv, gamma=0.995, lam=0.9):
    """Accumulated magnitude of the terms of the GAE recursion: m_t = |r_t| + gamma |V_{t+1}| + |V_t| + gamma lambda m_{t+1}.
    A few fp32 ulp of THIS is the attainable accuracy (the advantage itself can be a small difference of large terms)."""
    r, v, nnt = np.abs(np.asarray(r, np.float64)), np.abs(np.asarray(v, np.float64)), 1.0 - (np.asarray(d) != 0)
    m, last, nv = np.zeros_like(r), np.zeros(r.shape[1]), np.abs(np.asarray(lv, np.float64))
    for t in reversed(range(r.shape[0])):
        last = r[t] + gamma * nv * nnt[t] + v[t] + gamma * lam * nnt[t] * last
        m[t] = last
        nv = v[t]
    return m


def fixture_policy(device):
    z = np.load(os.path.join(GOLD, "policy_2300000.npz"))
    return ActorCritic.from_sb3({k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("_")}, device=device)


@pytest.mark.parametrize("T,N", [(1, 1), (37, 1000), (128, 4096), (5, 70001)])
def test_gae_kernel_matches_sb3_restatement(T, N):
    from oracle import oracle as O
    rng = np.random.RandomState(T * 131 + N)
    r = rng.randn(T, N).astype(np.float32) * 5
    v = rng.randn(T, N).astype(np.float32) * 50
    d = (rng.rand(T, N) < 0.05).astype(np.uint8)
    lv = rng.randn(N).astype(np.float32) * 50
    b = RolloutBuffer(T, N, 20, 4, "cuda")
    b.rewards.copy_(torch.from_numpy(r)); b.values.copy_(torch.from_numpy(v)); b.dones.copy_(torch.from_numpy(d)); b.last_values.copy_(torch.from_numpy(lv))
    adv, ret = compute_gae(b, 0.995, 0.9)
    adv_ref, ret_ref = O.gae_reference(r, v, d, lv, 0.995, 0.9)
    mag = gae_magnitude(r, v, d, lv)
    assert (np.abs(adv.cpu().numpy() - adv_ref) / (1.0 + mag)).max() < 1e-6
    assert (np.abs(ret.cpu().numpy() - ret_ref) / (1.0 + mag)).max() < 1e-6
    # where an episode ended, nothing from later steps leaks in: A_t = r_t - V_t exactly
    m = torch.from_numpy(d != 0).cuda()
    assert torch.equal(adv[m], (b.rewards - b.values)[m])


@pytest.mark.parametrize("A", [4, 7])
def test_gaussian_act_kernel(A):
    n = 200000
    dev = "cuda"
    mean = torch.randn(n, A, device=dev)
    log_std = torch.linspace(-1.0, 0.3, A, device=dev)
    low = torch.tensor([0.0] + [-1.0] * (A - 1), device=dev)
    high = torch.tensor([2.0] + [1.0] * (A - 1), device=dev)
    raw, clipped, logp = torch.zeros(n, A, device=dev), torch.zeros(n, A, device=dev), torch.zeros(n, device=dev)
    gaussian_act(mean, log_std, low, high, raw, clipped, logp, seed=7, draw=3)
    z = ((raw - mean) * torch.exp(-log_std)).double()
    # standard normal noise: moments per action dimension, independence across dimensions, tails
    assert float(z.mean(0).abs().max()) < 0.01 and float((z.var(0) - 1).abs().max()) < 0.02
    assert float(((z ** 4).mean(0) - 3).abs().max()) < 0.1
    c = torch.corrcoef(z.T) - torch.eye(A, device=dev, dtype=torch.float64)
    assert float(c.abs().max()) < 0.01
    assert 4.0 < float(z.abs().max()) < 6.5
    # log-prob is the diagonal-Gaussian density of the raw action; clip is the action-space clip
    ref = torch.distributions.Normal(mean, log_std.exp()).log_prob(raw).sum(-1)
    assert float((logp - ref).abs().max()) < 2e-4        # (a-mu)/sigma re-derived from a rounded a: ~1e-5 relative on z
    assert torch.equal(clipped, torch.minimum(torch.maximum(raw, low), high))
    # another draw index -> new noise; same (seed, global env id, draw) -> same noise however the envs are sharded
    raw2, c2, l2 = torch.zeros_like(raw), torch.zeros_like(raw), torch.zeros_like(logp)
    gaussian_act(mean, log_std, low, high, raw2, c2, l2, seed=7, draw=4)
    assert float((raw2 - raw).abs().mean()) > 0.1
    h = n // 2
    ra, ca, la = torch.zeros(h, A, device=dev), torch.zeros(h, A, device=dev), torch.zeros(h, device=dev)
    gaussian_act(mean[h:].contiguous(), log_std, low, high, ra, ca, la, seed=7, draw=3, env_id_offset=h)
    assert torch.equal(ra, raw[h:]) and torch.equal(la, logp[h:])


def test_reference_policy_flies_the_gpu_env():
    """The checkpoint the reference's README names as best (v2/README.md:55) in closed loop on the HIP environment: >95 %
    of episodes end in success, with the return and length the CPU oracle gives for the same policy (tests/test_ppo_cpu.py)."""
    env = amd.GpuWaypointEnv(2048, seed=3)
    pol = fixture_policy(env.device)
    obs = env.reset()
    env.stats(reset=True)
    for _ in range(1700):
        obs, _, _, _ = env.step(pol.predict(obs))
    s = env.stats()
    assert s["episodes"] > 3000
    assert s["success"] / s["episodes"] > 0.95
    assert 12000 < s["return_sum"] / s["episodes"] < 26000
    assert 500 < s["length_sum"] / s["episodes"] < 900


def test_rollout_buffer_is_consistent_with_policy_and_env():
    from oracle import oracle as O
    env = amd.GpuWaypointEnv(512, seed=11, max_episode_steps=40)           # short time limit: truncations inside the rollout
    algo = PPO(env, policy=fixture_policy(env.device), n_steps=96, batch_size=4096, n_epochs=1, seed=5)
    with torch.no_grad():
        algo.policy.log_std.data.fill_(-1.5)
    b = algo.collect_rollouts()
    T = algo.n_steps
    n = T * env.num_envs
    with torch.no_grad():
        values, logp, _ = algo.policy.evaluate_actions(b.obs[:T].reshape(n, -1), b.actions.reshape(n, -1))
    assert float((logp - b.logp.reshape(n)).abs().max()) < 2e-4             # ratio == 1 at the first minibatch, as in SB3
    assert float((values - b.values.reshape(n)).abs().max()) < 1e-3 * max(1.0, float(values.abs().max()))
    d = b.dones.bool()
    assert 0 < int(d.sum()) < n // 4
    # the time limit fires at step 41 of an episode: rewards there carry gamma * V(terminal_observation)
    assert int(d[40].sum()) > 400
    adv_ref, ret_ref = O.gae_reference(b.rewards.cpu().numpy(), b.values.cpu().numpy(), b.dones.cpu().numpy(), b.last_values.cpu().numpy(), 0.995, 0.9)
    mag = gae_magnitude(b.rewards.cpu().numpy(), b.values.cpu().numpy(), b.dones.cpu().numpy(), b.last_values.cpu().numpy())
    assert (np.abs(b.advantages.cpu().numpy() - adv_ref) / (1.0 + mag)).max() < 1e-6
    # second rollout continues from the last observation of the first
    last = b.obs[T].clone()
    b2 = algo.collect_rollouts()
    assert torch.equal(b2.obs[0], last)
    assert algo.num_timesteps == 2 * n


def test_truncation_bootstrap_matches_sb3_rule():
    """reward += gamma * V(terminal_observation) only where the episode was cut by the time limit (SB3 collect_rollouts)."""
    def rollout(bootstrap):
        env = amd.GpuWaypointEnv(256, seed=2, max_episode_steps=10)
        algo = PPO(env, policy=fixture_policy(env.device), n_steps=11, seed=1, bootstrap_truncated=bootstrap)
        return env, algo, algo.collect_rollouts()
    _, _, b0 = rollout(False)
    env, algo, b1 = rollout(True)
    assert torch.equal(b0.actions, b1.actions) and torch.equal(b0.dones, b1.dones)
    diff = b1.rewards - b0.rewards
    trunc = ((env.info_bits & 3) == L.INFO_TRUNCATED) & b1.dones[10].bool()     # the 11th step is past the 10-step limit
    assert int(trunc.sum()) > 200
    assert float(diff[:10].abs().max()) == 0.0 and float(diff[10][~trunc].abs().max() if bool((~trunc).any()) else 0.0) == 0.0
    with torch.no_grad():
        v = algo.policy.critic(env.terminal_obs)[trunc]                          # still in the env's buffer after the last step
    assert torch.allclose(diff[10][trunc], 0.995 * v, rtol=1e-4, atol=1e-3)
    assert float(v.abs().min()) > 0.0


def test_learn_runs():
    env = amd.GpuWaypointEnv(1024, seed=0)
    algo = PPO(env, n_steps=64, batch_size=8192, n_epochs=4, seed=0)
    p0 = algo.policy.flat_param.detach().clone()
    algo.learn(3 * 64 * 1024)
    assert len(algo.log) == 3 and algo.num_timesteps == 3 * 64 * 1024
    assert all(math.isfinite(v) for rec in algo.log for v in rec.values())
    assert float((algo.policy.flat_param.detach() - p0).abs().max()) > 1e-4
    assert all(rec["grad_norm"] > 0 and rec["episodes"] >= 0 for rec in algo.log)
    assert algo.log[0]["clip_fraction"] < 0.5


def test_fine_tuning_the_reference_checkpoint_keeps_it_flying(tmp_path):
    """`PPO.load(CHECKPOINT_PATH, env=env, ...)` + `learn` (v2/rl_train.py:33-35,56): resume from the reference's weights,
    train briefly with the reference's hyper-parameters (scaled batch), save, reload -- the policy still succeeds."""
    env = amd.GpuWaypointEnv(2048, seed=21)
    algo = PPO(env, n_steps=128, batch_size=16384, n_epochs=2, ent_coef=1e-4, seed=3)
    z = np.load(os.path.join(GOLD, "policy_2300000.npz"))
    algo.load_policy({k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("_")})
    algo.learn(2 * 128 * 2048)
    path = algo.save(os.path.join(tmp_path, "ppo_model_gpu"))
    pol = ActorCritic.from_sb3(path, device=env.device)
    assert torch.equal(pol.flatten_().flat_param, algo.policy.flat_param.detach())
    ev = amd.GpuWaypointEnv(1024, seed=99)
    obs = ev.reset()
    ev.stats(reset=True)
    for _ in range(1700):
        obs, _, _, _ = ev.step(pol.predict(obs))
    s = ev.stats()
    assert s["episodes"] > 1500 and s["success"] / s["episodes"] > 0.9


def test_ppo_with_arm_env_and_obs_normalizer():
    env = amd.GpuWaypointEnv(512, vehicle="hexa_arm", seed=4)
    norm = amd.ObsNormalizer(env.obs_dim, device=env.device_index)
    algo = PPO(env, n_steps=32, batch_size=4096, n_epochs=2, seed=0, obs_normalizer=norm)
    algo.learn(2 * 32 * 512)
    assert algo.policy.num_parameters() == 2 * (29 * 128 + 128 + 128 * 64 + 64 + 64 * 64 + 64) + 64 * 7 + 7 + 64 + 1 + 7
    assert all(math.isfinite(v) for rec in algo.log for v in rec.values())
    mean, var, count = norm.get()
    assert abs(count - (2 * 32 + 1) * 512) < 1.0
    assert float(algo.buffer.obs.abs().max()) <= 10.0                         # clip_obs


def test_graph_captured_update_equals_eager_update():
    """The HIP-graph replay of the minibatch step is the same arithmetic as the eager step: same rollout, same shuffles ->
    the same parameters (up to GEMM kernel selection) after 2 epochs x 8 minibatches, on one GPU."""
    out = []
    for use_graph in (False, True):
        env = amd.GpuWaypointEnv(1024, seed=8)
        algo = PPO(env, policy=fixture_policy(env.device), n_steps=64, batch_size=8192, n_epochs=2, seed=4, use_graph=use_graph)
        algo.collect_rollouts()
        rec = algo.train()
        out.append((algo.policy.flat_param.detach().clone(), rec, algo._step._graphs is not None))
    (p0, r0, g0), (p1, r1, g1) = out
    assert not g0 and g1                                                     # the second run really replayed graphs
    assert float((p0 - p1).abs().max()) < 1e-5
    assert abs(r0["value_loss"] - r1["value_loss"]) <= 1e-4 * abs(r0["value_loss"]) and abs(r0["grad_norm"] - r1["grad_norm"]) <= 1e-3 * r0["grad_norm"]


def test_split_graph_update_with_rccl_all_reduce_between():
    """The multi-GPU shape of the update on one GPU: a 1-rank RCCL (nccl backend) process group, forward/backward graph ->
    eager all-reduce of the flat gradient buffer -> clip + Adam graph.  With one rank the all-reduce is the identity, so the
    result must equal the single-graph update."""
    import socket
    import torch.distributed as dist
    from rl_aerial_manipulator_amd.ppo import MinibatchStep
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        out = []
        for split in (False, True):
            env = amd.GpuWaypointEnv(1024, seed=8)
            algo = PPO(env, policy=fixture_policy(env.device), n_steps=64, batch_size=8192, n_epochs=2, seed=4)
            algo._step = MinibatchStep(algo.policy, algo.optimizer, ent_coef=algo.ent_coef, dist=dist, use_graph=True, split_graphs=split)
            algo.collect_rollouts()
            algo.train()
            out.append((algo.policy.flat_param.detach().clone(), algo._step._graphs))
        t = torch.ones(4, device="cuda")
        dist.all_reduce(t)
        dist.barrier()
        assert out[0][1][1] is None and out[1][1][1] is not None          # one graph vs two
        assert float((out[0][0] - out[1][0]).abs().max()) < 1e-6 and float(t.sum()) == 4.0
    finally:
        dist.destroy_process_group()


def test_evaluate_policy_on_gpu():
    """`evaluate_policy(model, env, n_eval_episodes=10)` (v2/rl_train.py:60) with the reference checkpoint on the HIP env."""
    from rl_aerial_manipulator_amd.ppo import evaluate_policy
    env = amd.GpuWaypointEnv(512, seed=12)
    mean, std = evaluate_policy(fixture_policy(env.device), env, n_eval_episodes=1024)
    assert 12000 < mean < 26000 and 0 < std < 12000


@pytest.mark.parametrize("A,normalize", [(4, True), (7, True), (4, False)])
def test_fused_ppo_loss_kernel_matches_autograd(A, normalize):
    """amenv_ppo_loss_grad (advantage normalisation, log-prob, ratio, clipped surrogate, value MSE, entropy and all their
    gradients in three launches) against the torch expression of SB3's loss differentiated by autograd, in fp64."""
    from rl_aerial_manipulator_amd.ppo import MinibatchStep
    torch.manual_seed(A)
    n, D = 10007, 20 if A == 4 else 29
    dev = "cuda"
    pol = ActorCritic(D, A).to(dev).flatten_()
    with torch.no_grad():
        pol.log_std.data.copy_(torch.linspace(-0.7, 0.2, A))
    obs = torch.randn(n, D, device=dev)
    with torch.no_grad():
        mean = pol.actor(obs)
    actions = (mean + torch.randn(n, A, device=dev) * pol.log_std.detach().exp() * 1.5).contiguous()
    old_logp = (pol.evaluate_actions(obs, actions)[1].detach() + 0.3 * torch.randn(n, device=dev)).contiguous()   # ratios well outside the clip range too
    adv = (torch.randn(n, device=dev) * 3 + 0.5).contiguous()
    ret = (torch.randn(n, device=dev) * 10).contiguous()
    leaf = pol.flat_param.requires_grad_(True)
    opt = torch.optim.SGD([leaf], lr=0.0)
    fused = MinibatchStep(pol, opt, normalize_advantage=normalize, use_graph=False, fused_loss=True)
    fused._forward_backward(obs, actions, old_logp, adv, ret)
    g_fused, s_fused = pol.flat_grad.clone(), fused.stats.clone()
    # reference: the same loss in fp64 through autograd
    pol64 = ActorCritic(D, A).to(dev).double()
    pol64.load_state_dict({k: v.double() for k, v in pol.state_dict().items()})
    a64 = adv.double()
    if normalize:
        a64 = (a64 - a64.mean()) / (a64.std() + 1e-8)
    values, logp, ent = pol64.evaluate_actions(obs.double(), actions.double())
    ratio = torch.exp(logp - old_logp.double())
    pl = -torch.min(a64 * ratio, a64 * ratio.clamp(0.8, 1.2)).mean()
    vl = ((ret.double() - values) ** 2).mean()
    el = -ent.mean()
    grads = torch.autograd.grad(pl + 5e-4 * el + 0.5 * vl, list(pol64.parameters()))
    g_ref = torch.cat([g.reshape(-1) for g in grads])
    scale = g_ref.abs().max()
    assert float((g_fused.double() - g_ref).abs().max() / scale) < 2e-5
    ref_stats = torch.stack([pl, vl, el, ((ratio - 1).abs() > 0.2).double().mean()])
    assert float(((s_fused[:4].double() - ref_stats).abs() / ref_stats.abs().clamp(min=1e-3)).max()) < 1e-4
    assert 0.05 < float(ref_stats[3]) < 0.95                                   # both clipped and unclipped samples present
    # and the eager torch path of the same class agrees (it is what the CPU / gloo tests exercise)
    plain = MinibatchStep(pol, opt, normalize_advantage=normalize, use_graph=False, fused_loss=False)
    plain._forward_backward(obs, actions, old_logp, adv, ret)
    assert float((pol.flat_grad.double() - g_ref).abs().max() / scale) < 2e-5


@pytest.mark.parametrize("D,A", [(20, 4), (29, 7), (17, 4)])
@pytest.mark.parametrize("n", [1, 64, 1000, 10007, 32768])
def test_fused_policy_forward_matches_torch_modules(D, A, n):
    """amenv_policy_forward (one launch: scalar-operand weights, LDS activations; n < 8192) and amenv_policy_forward_mfma (the training
    kernel's forward passes on the matrix cores; larger n, ragged last tile included) against the torch modules they replace on the
    inference path, for the three (obs, action) shapes of the envs."""
    torch.manual_seed(D * 100 + A)
    pol = ActorCritic(D, A).to("cuda").flatten_()
    with torch.no_grad():
        pol.flat_param.mul_(1.7)                                    # larger pre-activations: tanh away from its linear range
        pol.action_net.weight.mul_(30.0)
    obs = torch.randn(n, D, device="cuda") * 1.5
    with torch.no_grad():
        assert pol.fused_ok(obs)
        mean, value = pol.actor_critic(obs)
        ref_mean = pol.action_net(pol.mlp_extractor.policy_net(obs))
        ref_value = pol.value_net(pol.mlp_extractor.value_net(obs)).squeeze(-1)
        only_v = pol.critic(obs)
    assert mean.shape == (n, A) and value.shape == (n,)
    assert float((mean - ref_mean).abs().max()) < 2e-5 * max(1.0, float(ref_mean.abs().max()))
    assert float((value - ref_value).abs().max()) < 2e-5 * max(1.0, float(ref_value.abs().max()))
    assert torch.equal(only_v, value)
    with torch.enable_grad():                                           # training passes stay on the torch modules
        assert not pol.fused_ok(obs) and pol.actor(obs).requires_grad


def test_fused_policy_forward_reproduces_recorded_actions_and_is_capturable():
    """The reference checkpoint through the fused forward: the actions recorded from the reference (golden policy episodes), and the
    same launch replayed from a HIP graph."""
    import glob
    pol = fixture_policy("cuda")
    n = 0
    for f in sorted(glob.glob(os.path.join(GOLD, "policy_ep*.npz"))):
        g = np.load(f)
        obs_seen = torch.from_numpy(np.concatenate([g["obs0"][None], g["obs"][:-1]])).cuda()
        a = pol.predict(obs_seen)
        assert pol.flat_param is not None and float((a.cpu() - torch.from_numpy(g["actions"])).abs().max()) < 3e-6
        n += len(a)
    assert n > 2000
    obs = torch.randn(4096, 20, device="cuda")
    out = pol.predict(obs)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out2 = pol.predict(obs)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, out2)


def test_matrix_core_forward_reproduces_the_recorded_reference_actions():
    """amenv_policy_forward_mfma (batches >= 8192 rows) on the reference checkpoint: every recorded observation of the four golden policy
    episodes, tiled past the switch-over, gives the action the reference recorded (to 6e-6; the small-batch kernel's bar is 3e-6) and the
    small-batch kernel's means and values to the same tolerance."""
    import glob
    pol = fixture_policy("cuda")
    if pol.flat_param is None:
        pol.flatten_()
    obs_l, act_l = [], []
    for f in sorted(glob.glob(os.path.join(GOLD, "policy_ep*.npz"))):
        g = np.load(f)
        obs_l.append(np.concatenate([g["obs0"][None], g["obs"][:-1]])); act_l.append(g["actions"])
    obs = torch.from_numpy(np.concatenate(obs_l)).cuda()
    rec = torch.from_numpy(np.concatenate(act_l)).cuda()
    n = obs.shape[0]
    reps = -(-pol.MFMA_FORWARD_ROWS // n) + 1
    big = obs.repeat(reps, 1)[: reps * n - 5].contiguous()                       # ragged last tile
    assert big.shape[0] >= pol.MFMA_FORWARD_ROWS
    with torch.no_grad():
        small_mean, small_value = pol.forward_fused(obs)                       # VALU kernel
        big_mean, big_value = pol.forward_fused(big)                           # matrix cores
    lo, hi = pol.action_low, pol.action_high
    a = torch.minimum(torch.maximum(big_mean[:n], lo), hi)
    err = float((a - rec).abs().max())
    print(f"matrix-core forward vs the {n} recorded actions: max |difference| {err:.2e}")
    assert err < 6e-6                                                            # (the small-batch kernel: 3e-6; two fp32 summation orders of a trained, stiff controller)
    assert float((big_mean[:n] - small_mean).abs().max()) < 6e-6 * max(1.0, float(small_mean.abs().max()))
    assert float((big_value[:n] - small_value).abs().max()) < 6e-6 * max(1.0, float(small_value.abs().max()))
    assert torch.equal(big_mean[n:2 * n], big_mean[:n])                          # tiles are independent of their position


def test_training_script_runs(tmp_path):
    """examples/rl_train_gpu.py (the reference's rl_train.py on the GPU stack) end to end for two small iterations: trains, saves
    a checkpoint whose policy.pth has SB3's layout, evaluates."""
    import subprocess
    import sys
    out = os.path.join(tmp_path, "model")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "rl_train_gpu.py"), "--envs", "128", "--n-steps", "64", "--timesteps", "16384",
                        "--save", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Mean reward:" in r.stdout and os.path.exists(out + ".zip")
    pol = ActorCritic.from_sb3(out + ".zip")
    assert pol.num_parameters() == 30537


@pytest.mark.parametrize("vehicle", ["quad", "hexa_arm"])
def test_ppo_at_config5_batch_32768_envs(vehicle):
    """BASELINE configs[4]: PPO rollout + update at 32768 envs on ONE GPU (the per-GPU share of the 8-GPU job).  Short rollouts, 2
    iterations; the invariants asserted at 512..2048 envs hold at this size: buffer consistent with policy and env, GAE equal to the
    SB3 restatement, finite losses, parameters move, and the HIP-graph update equals the eager update on the same rollout."""
    from oracle import oracle as O
    n, T = 32768, 8
    out = []
    for use_graph in (False, True):
        env = amd.GpuWaypointEnv(n, vehicle=vehicle, seed=17, max_episode_steps=6)     # truncations inside the short rollout
        assert env.num_envs == n
        algo = PPO(env, n_steps=T, batch_size=65536, n_epochs=2, seed=9, use_graph=use_graph)
        with torch.no_grad():
            algo.policy.log_std.data.fill_(-1.0)
        b = algo.collect_rollouts()
        N = T * n
        with torch.no_grad():
            values, logp, _ = algo.policy.evaluate_actions(b.obs[:T].reshape(N, -1), b.actions.reshape(N, -1))
        assert float((logp - b.logp.reshape(N)).abs().max()) < 5e-4
        assert float((values - b.values.reshape(N)).abs().max()) < 1e-3 * max(1.0, float(values.abs().max()))
        d = b.dones.bool()
        assert int(d[6].sum()) > n // 2                                               # the 7th step is past the 6-step limit
        if not use_graph:
            r, v, dn, lv = (x.cpu().numpy() for x in (b.rewards, b.values, b.dones, b.last_values))
            adv_ref, _ = O.gae_reference(r, v, dn, lv, 0.995, 0.9)
            assert (np.abs(b.advantages.cpu().numpy() - adv_ref) / (1.0 + gae_magnitude(r, v, dn, lv))).max() < 1e-6
        p0 = algo.policy.flat_param.detach().clone()
        rec = algo.train()
        assert all(math.isfinite(x) for x in rec.values()) and rec["grad_norm"] > 0
        assert float((algo.policy.flat_param.detach() - p0).abs().max()) > 1e-5
        out.append((algo.policy.flat_param.detach().clone(), rec, algo._step._graphs is not None))
        algo.learn(N)                                                                  # one more full iteration through learn()
        assert algo.num_timesteps == 2 * N and math.isfinite(algo.log[-1]["value_loss"])
        env.close()
    (pa, ra, ga), (pb, rb, gb) = out
    assert not ga and gb
    assert float((pa - pb).abs().max()) < 2e-5 and abs(ra["value_loss"] - rb["value_loss"]) <= 1e-4 * abs(ra["value_loss"])


def test_checkpoint_cadence_and_full_resume(tmp_path):
    """CheckpointCallback(save_freq, save_path, name_prefix) + PPO.load(...) then learn (v2/rl_train.py:14-18,33-35,56): checkpoints land
    at the cadence, and a resumed run restores weights, Adam moments, step count, learning rate and the timestep counter."""
    env = amd.GpuWaypointEnv(512, seed=1)
    algo = PPO(env, n_steps=32, batch_size=4096, n_epochs=2, seed=2, learning_rate=3e-4)
    per_iter = 32 * 512
    algo.learn(5 * per_iter, save_freq=2 * per_iter, save_path=str(tmp_path), name_prefix="ppo_model")
    files = sorted(os.listdir(tmp_path))
    assert files == [f"ppo_model_{2 * per_iter}_steps.zip", f"ppo_model_{4 * per_iter}_steps.zip"], files
    final = algo.save(os.path.join(tmp_path, "final"))
    st = algo.optimizer.state[algo._leaf]
    env2 = amd.GpuWaypointEnv(512, seed=1)
    res = PPO(env2, n_steps=32, batch_size=4096, n_epochs=2, seed=2)            # default lr 2e-4: must be overwritten by the checkpoint
    res.load(final)
    st2 = res.optimizer.state[res._leaf]
    assert torch.equal(res.policy.flat_param.detach(), algo.policy.flat_param.detach())
    assert float(st2["step"]) == float(st["step"]) == 5 * 2 * 4 and torch.equal(st2["exp_avg"], st["exp_avg"]) and torch.equal(st2["exp_avg_sq"], st["exp_avg_sq"])
    assert res.optimizer.param_groups[0]["lr"] == 3e-4 and res.num_timesteps == 5 * per_iter and res._draw == algo._draw
    res.learn(per_iter)
    assert res.num_timesteps == 6 * per_iter and all(math.isfinite(v) for v in res.log[-1].values())
    assert float(res.optimizer.state[res._leaf]["step"]) == 6 * 2 * 4
    env.close(); env2.close()


def test_fused_rollout_feeds_the_update():
    """PPO(fused_rollout=True): collect_rollouts is ONE launch of amenv_rollout_policy (bf16 policy on the matrix cores + env steps); the
    buffer it fills is consistent with the fp32 policy the update differentiates (values / log-probs within bf16 tolerance, ratio ~ 1
    at the first minibatch), GAE matches the SB3 restatement, learn() runs and moves the parameters."""
    from oracle import oracle as O
    env = amd.GpuWaypointEnv(2048, vehicle="hexa_arm", seed=5, max_episode_steps=30)      # truncations inside the rollout
    algo = PPO(env, n_steps=64, batch_size=16384, n_epochs=2, seed=3, fused_rollout=True)
    with torch.no_grad():
        algo.policy.log_std.data.fill_(-1.0)
    b = algo.collect_rollouts()
    T, n = 64, 2048
    N = T * n
    with torch.no_grad():
        values, logp, _ = algo.policy.evaluate_actions(b.obs[:T].reshape(N, -1), b.actions.reshape(N, -1))
    assert float((values - b.values.reshape(N)).abs().max()) < 3e-2 * max(1.0, float(values.abs().max()))
    assert float((logp - b.logp.reshape(N)).abs().mean()) < 0.05 and float((logp - b.logp.reshape(N)).abs().max()) < 0.6
    d = b.dones.bool()
    assert int(d[30].sum()) > n // 2 and algo._draw == T
    # rewards at truncated entries carry gamma * V(terminal_observation); GAE over the buffer as SB3 computes it
    r, v, dn, lv = (x.cpu().numpy() for x in (b.rewards, b.values, b.dones, b.last_values))
    adv_ref, _ = O.gae_reference(r, v, dn, lv, 0.995, 0.9)
    assert (np.abs(b.advantages.cpu().numpy() - adv_ref) / (1.0 + gae_magnitude(r, v, dn, lv))).max() < 1e-6
    p0 = algo.policy.flat_param.detach().clone()
    rec = algo.train()
    assert all(math.isfinite(x) for x in rec.values()) and rec["clip_fraction"] < 0.5
    algo.learn(2 * N)
    assert algo.num_timesteps == 3 * N and float((algo.policy.flat_param.detach() - p0).abs().max()) > 1e-5
    assert all(math.isfinite(x) for rec in algo.log for x in rec.values())
    env.close()


@pytest.mark.parametrize("D,A,n", [(29, 7, 8192), (20, 4, 5000), (17, 4, 31), (29, 7, 65536), (20, 4, 20011)])   # 20011: two tile rounds, the second ragged
def test_fused_mlp_step_matches_autograd(D, A, n):
    """amenv_ppo_mlp_step (forward + SB3 loss + backward + all weight gradients of both MLPs in one kernel; fp32 products, formed since round
    3 from six bf16 MFMAs on exactly split operands -- the gate below is the fp32-MFMA version's, unchanged) against autograd on the fp32 torch modules with the torch statement of the loss: every gradient entry within 2e-5 of the largest,
    the four reported scalars equal; ragged batch sizes (n not a multiple of 32) included."""
    from rl_aerial_manipulator_amd.ppo import MinibatchStep
    torch.manual_seed(3)
    pol = ActorCritic(D, A).cuda().flatten_()
    with torch.no_grad():
        pol.log_std.data.copy_(torch.linspace(-0.7, 0.2, A))
        pol.action_net.weight.mul_(20.0)
    opt = torch.optim.Adam([pol.flat_param.requires_grad_(True)], lr=1e-3)
    g = torch.Generator(device="cuda").manual_seed(1)
    obs = torch.randn(n, D, device="cuda", generator=g) * 0.7
    with torch.no_grad():
        mean = pol.action_net(pol.mlp_extractor.policy_net(obs))
    actions = mean + torch.exp(pol.log_std.detach()) * torch.randn(n, A, device="cuda", generator=g)
    with torch.no_grad():
        _, logp, _ = pol.evaluate_actions(obs, actions)
    old_logp = logp + 0.15 * torch.randn(n, device="cuda", generator=g)      # ratios spread around 1: both sides of the clip fire
    adv = torch.randn(n, device="cuda", generator=g) * 3.0 + 0.5
    ret = torch.randn(n, device="cuda", generator=g) * 2.0
    outs = []
    for fused in (False, True):
        step = MinibatchStep(pol, opt, clip_range=0.2, ent_coef=5e-4, vf_coef=0.5, use_graph=False, fused_loss=False, fused_mlp=fused)
        assert step.fused_mlp == fused
        pol.flat_grad.zero_()
        step._forward_backward(obs, actions, old_logp, adv, ret)
        torch.cuda.synchronize()
        outs.append((pol.flat_grad.clone(), step.stats[:4].clone()))
    (g0, s0), (g1, s1) = outs
    scale = float(g0.abs().max())
    assert scale > 0 and float((g0 - g1).abs().max()) < 2e-5 * scale, (float((g0 - g1).abs().max()), scale)
    # per parameter block as well: small blocks (biases, log_std) are not hidden behind the largest entry
    off = 0
    for p_ in pol.parameters():
        k = p_.numel()
        blk = float(g0[off:off + k].abs().max())
        assert float((g0[off:off + k] - g1[off:off + k]).abs().max()) < 1e-4 * max(blk, 1e-3 * scale), (off, k)
        off += k
    assert torch.allclose(s0, s1, rtol=2e-4, atol=1e-6), (s0, s1)
    assert 0.02 < float(s1[3]) < 0.9                                       # clip fraction: the clipped branch was exercised


@pytest.mark.parametrize("obs_scale,w_scale", [(0.7, 1.0), (0.02, 30.0), (6.0, 0.2)])
def test_fused_mlp_step_is_as_accurate_as_fp32_autograd(obs_scale, w_scale):
    """The split-bf16 products are fp32-grade: against the SAME loss differentiated in fp64, the fused kernel's gradient error is no larger
    than a few times that of torch's own fp32 modules -- also with small inputs x large first-layer weights and the other way round (the three
    bf16 parts carry 24 significant bits whatever the magnitude; a two-part split would be ~100 x worse and fail this)."""
    from rl_aerial_manipulator_amd.ppo import MinibatchStep
    D, A, n = 29, 7, 16384
    torch.manual_seed(11)
    pol = ActorCritic(D, A).cuda().flatten_()
    with torch.no_grad():
        pol.log_std.data.copy_(torch.linspace(-0.5, 0.1, A))
        pol.mlp_extractor.policy_net[0].weight.mul_(w_scale)
        pol.mlp_extractor.value_net[0].weight.mul_(w_scale)
    opt = torch.optim.Adam([pol.flat_param.requires_grad_(True)], lr=1e-3)
    g = torch.Generator(device="cuda").manual_seed(2)
    obs = torch.randn(n, D, device="cuda", generator=g) * obs_scale
    with torch.no_grad():
        mean = pol.action_net(pol.mlp_extractor.policy_net(obs))
    actions = mean + torch.exp(pol.log_std.detach()) * torch.randn(n, A, device="cuda", generator=g)
    with torch.no_grad():
        _, logp, _ = pol.evaluate_actions(obs, actions)
    old_logp = logp + 0.15 * torch.randn(n, device="cuda", generator=g)
    adv = torch.randn(n, device="cuda", generator=g) * 3.0 + 0.5
    ret = torch.randn(n, device="cuda", generator=g) * 2.0
    grads = {}
    for fused in (False, True):
        step = MinibatchStep(pol, opt, clip_range=0.2, ent_coef=5e-4, vf_coef=0.5, use_graph=False, fused_loss=False, fused_mlp=fused)
        assert step.fused_mlp == fused
        pol.flat_grad.zero_()
        step._forward_backward(obs, actions, old_logp, adv, ret)
        grads[fused] = pol.flat_grad.double().clone()
    pol64 = ActorCritic(D, A).cuda().double()
    pol64.load_state_dict({k: v.double() for k, v in pol.state_dict().items() if k in pol64.state_dict()})
    a64 = adv.double()
    a64 = (a64 - a64.mean()) / (a64.std() + 1e-8)
    values, logp64, ent = pol64.evaluate_actions(obs.double(), actions.double())
    ratio = torch.exp(logp64 - old_logp.double())
    loss = -torch.min(a64 * ratio, a64 * ratio.clamp(0.8, 1.2)).mean() - 5e-4 * ent.mean() + 0.5 * ((ret.double() - values) ** 2).mean()
    g64 = torch.cat([x.reshape(-1) for x in torch.autograd.grad(loss, list(pol64.parameters()))])
    scale = float(g64.abs().max())
    e_torch, e_fused = float((grads[False] - g64).abs().max()) / scale, float((grads[True] - g64).abs().max()) / scale
    print(f"obs x{obs_scale} W1 x{w_scale}: relative error vs fp64 autograd: torch fp32 {e_torch:.2e}, fused kernel {e_fused:.2e}")
    assert e_fused < max(4.0 * e_torch, 2e-6), (e_fused, e_torch)


def _mlp_problem(D, A, n, seed=1):
    torch.manual_seed(3)
    pol = ActorCritic(D, A).cuda().flatten_()
    g = torch.Generator(device="cuda").manual_seed(seed)
    obs = torch.randn(n, D, device="cuda", generator=g) * 0.7
    actions = torch.randn(n, A, device="cuda", generator=g) * 0.5
    with torch.no_grad():
        _, logp, _ = pol.evaluate_actions(obs, actions)
    old_logp = logp + 0.15 * torch.randn(n, device="cuda", generator=g)
    adv = torch.randn(n, device="cuda", generator=g) * 3.0 + 0.5
    ret = torch.randn(n, device="cuda", generator=g) * 2.0
    return pol, (obs, actions, old_logp, adv, ret)


def test_fused_mlp_step_index_gather_is_bit_identical_and_deterministic():
    """`index` makes the kernel read rows index[s] of the rollout tensors: the gradient equals, bit for bit, the one from the gathered
    copies; and the step has no atomics on the gradient path, so a repeated call gives the same bits."""
    from rl_aerial_manipulator_amd.ppo import MinibatchStep
    n_roll, n = 50000, 12345
    pol, full = _mlp_problem(29, 7, n_roll)
    opt = torch.optim.Adam([pol.flat_param.requires_grad_(True)], lr=1e-3, capturable=True)
    step = MinibatchStep(pol, opt, use_graph=False)
    assert step.fused_mlp
    idx = torch.randperm(n_roll, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))[:n]
    step._forward_backward_mlp(*full, idx)
    g_idx, s_idx = pol.flat_grad.clone(), step.stats[:4].clone()
    step._forward_backward_mlp(*[t[idx].contiguous() for t in full])
    g_cpy, s_cpy = pol.flat_grad.clone(), step.stats[:4].clone()
    assert torch.equal(g_idx, g_cpy) and torch.equal(s_idx, s_cpy)
    step._forward_backward_mlp(*full, idx)
    assert torch.equal(pol.flat_grad, g_idx)
    assert float(g_idx.abs().max()) > 0


@pytest.mark.parametrize("world_scale,max_norm,n", [(1, 0.5, 33039), (2, 0.5, 33039), (1, None, 33039), (1, 0.5, 300001)])
def test_fused_adam_step_matches_torch_adam(world_scale, max_norm, n):
    """amenv_ppo_adam_step against clip_grad_norm_-style scaling + torch.optim.Adam over several steps: parameters, both moments, the
    step counter and the reported norm; the gradient buffer is read-only for the kernel.  n = 33039: the reference policy's size, 300001: a
    larger buffer (every workgroup still sums all of it)."""
    from rl_aerial_manipulator_amd import _lib as L_
    import ctypes as C_
    torch.manual_seed(0)
    p_ref = torch.randn(n, device="cuda", dtype=torch.float64) * 0.3
    p_hip = p_ref.float().clone()
    leaf = p_ref.clone().requires_grad_(True)
    opt = torch.optim.Adam([leaf], lr=2e-4, eps=1e-5)
    m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda"); stp = torch.zeros((), device="cuda")
    hyper = torch.tensor([2e-4, 0.9, 0.999, 1e-5, max_norm or 0.0, 1.0 / world_scale], device="cuda")
    gn = torch.zeros(1, device="cuda"); ticket = torch.zeros(1, dtype=torch.int32, device="cuda")
    ptr = lambda t: C_.c_void_p(t.data_ptr())  # noqa: E731
    for it in range(25):
        g = torch.randn(n, device="cuda") * (0.05 if it % 3 else 0.001) * world_scale
        gd = g.double() / world_scale
        norm = gd.norm(2)
        if max_norm is not None:
            gd = gd * torch.clamp(max_norm / (norm + 1e-6), max=1.0)
        leaf.grad = gd.clone()
        opt.step()
        gh = g.clone()
        rc = L_.load().amenv_ppo_adam_step(ptr(p_hip), ptr(gh), ptr(m), ptr(v), ptr(stp), n, ptr(hyper), ptr(gn), ptr(ticket), None)
        assert rc == 0
        assert abs(float(gn) - float(norm)) < 1e-5 * float(norm)
        assert torch.equal(gh, g)                                                   # read-only: the clipped gradient is applied, not stored
    st = opt.state[leaf]
    assert float(stp) == 25.0 and int(ticket) == 0
    assert torch.allclose(m.double(), st["exp_avg"], rtol=1e-4, atol=1e-7)
    assert torch.allclose(v.double(), st["exp_avg_sq"], rtol=1e-4, atol=1e-12)
    # 25 Adam steps of at most lr each: the fp32 kernel tracks the fp64 optimiser to a small fraction of one step
    assert float((p_hip.double() - leaf.detach()).abs().max()) < 2e-4 * 0.02
    assert float((p_hip.double() - p_ref).abs().max()) > 1e-3


@pytest.mark.parametrize("n", [33039, 300001])
def test_fused_adam_step_norm_is_complete_before_any_gradient_is_clipped(n):
    """The clip factor comes from the norm of the WHOLE gradient as it was on entry: repeated from the same inputs with the clip active,
    every run gives the same bits and leaves the gradient buffer untouched.  (A first version overwrote the buffer with the clipped
    gradient while other workgroups of the launch were still summing it: found by the bit-exact RCCL-exchange test below failing once.)"""
    from rl_aerial_manipulator_amd import _lib as L_
    import ctypes as C_
    torch.manual_seed(1)
    g0 = torch.randn(n, device="cuda") * 0.3                          # norm >> 0.5: clip factor ~ 0.01
    p0 = torch.randn(n, device="cuda")
    hyper = torch.tensor([2e-4, 0.9, 0.999, 1e-5, 0.5, 1.0], device="cuda")
    word = torch.zeros(1, dtype=torch.int32, device="cuda")
    ptr = lambda t: C_.c_void_p(t.data_ptr())  # noqa: E731
    first = None
    for _ in range(200):
        p, g, m, v, stp, gn = p0.clone(), g0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros((), device="cuda"), torch.zeros(1, device="cuda")
        assert L_.load().amenv_ppo_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(stp), n, ptr(hyper), ptr(gn), ptr(word), None) == 0
        assert torch.equal(g, g0)
        if first is None:
            first = (p.clone(), m.clone(), float(gn))
            assert abs(first[2] - float(g0.double().norm())) < 1e-4 * first[2]
            # the applied gradient has norm max_grad_norm: exp_avg = (1 - beta1) * clipped gradient after the first step
            assert abs(float((m.double() / 0.1).norm()) - 0.5) < 1e-4
        else:
            assert torch.equal(p, first[0]) and torch.equal(m, first[1]) and float(gn) == first[2]


def test_ppo_update_fused_path_trains_like_the_torch_path():
    """`ppo_update` with the fused MLP + Adam kernels (indexed minibatches, no graph) against the same update through the torch modules
    and torch.optim.Adam, same permutations: parameters after 2 epochs x 4 minibatches agree to fp32 round-off of a few Adam steps."""
    from rl_aerial_manipulator_amd.ppo import MinibatchStep, ppo_update
    res = []
    for fused in (False, True):
        pol, data = _mlp_problem(29, 7, 16384)
        opt = torch.optim.Adam([pol.flat_param.requires_grad_(True)], lr=2e-4, eps=1e-5, capturable=True)
        pol.flat_param.grad = pol.flat_grad
        step = MinibatchStep(pol, opt, use_graph=False, fused_mlp=fused)
        assert step.fused_mlp == fused and step.fused_adam == fused
        gen = torch.Generator(device="cuda").manual_seed(11)
        out = ppo_update(pol, opt, *data, batch_size=4096, n_epochs=2, generator=gen, step=step)
        res.append((pol.flat_param.detach().clone(), out, float(opt.state[opt.param_groups[0]["params"][0]]["step"])))
    (p0, o0, t0), (p1, o1, t1) = res
    assert t0 == t1 == 8.0
    assert float((p0 - p1).abs().max()) < 0.05 * 8 * 2e-4, float((p0 - p1).abs().max())
    for k in o0:
        assert abs(o0[k] - o1[k]) <= 2e-3 * max(abs(o0[k]), 1e-3), (k, o0[k], o1[k])


def test_fused_update_with_rccl_exchange_between_reduce_and_adam():
    """The multi-GPU form of the fused update -- gradient reduce | RCCL all-reduce of the flat gradient | clip + Adam with the 1 / world
    scale folded in -- on a 1-rank RCCL group (forced exchange): same parameters, bit for bit, as the update without a process group."""
    import socket
    import torch.distributed as dist
    from rl_aerial_manipulator_amd.ppo import MinibatchStep, ppo_update
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        res = []
        for with_dist in (False, True):
            pol, data = _mlp_problem(29, 7, 16384)
            opt = torch.optim.Adam([pol.flat_param.requires_grad_(True)], lr=2e-4, eps=1e-5, capturable=True)
            pol.flat_param.grad = pol.flat_grad
            step = MinibatchStep(pol, opt, dist=dist if with_dist else None, split_graphs=True if with_dist else None)
            assert step.fused_mlp and step.fused_adam and not step.use_graph
            gen = torch.Generator(device="cuda").manual_seed(11)
            ppo_update(pol, opt, *data, batch_size=4096, n_epochs=2, generator=gen, step=step)
            res.append(pol.flat_param.detach().clone())
        t = torch.ones(4, device="cuda")
        dist.all_reduce(t)
        assert torch.equal(res[0], res[1]) and float(t.sum()) == 4.0
    finally:
        dist.destroy_process_group()


# ---- closed-loop rollout in one launch for the rigid vehicles (amenv_quad_policy.hpp) ----------------------------
def _quad_rollout_buffers(T, n, dev):
    return dict(obs=torch.zeros(T + 1, n, 20, device=dev), actions=torch.zeros(T, n, 4, device=dev), logp=torch.zeros(T, n, device=dev),
                values=torch.zeros(T, n, device=dev), rewards=torch.zeros(T, n, device=dev), dones=torch.zeros(T, n, dtype=torch.uint8, device=dev))


@pytest.mark.parametrize("vehicle,n", [("quad", 300), ("quad", 4096), ("hexa", 1000)])
def test_rigid_closed_loop_policy_rollout_kernel(vehicle, n):
    """amenv_rollout_policy on the reference's vehicle: the env part EXACTLY (replaying the recorded clipped actions through amenv_step on a
    lane-quad env reproduces every observation / reward / flag bit for bit), the policy part against the fp32 modules (bf16 tolerance), the
    noise statistically, determinism."""
    T = 96
    torch.manual_seed(7)
    pol = ActorCritic(20, 4).cuda().flatten_()
    with torch.no_grad():
        pol.log_std.data.fill_(-1.2)
        pol.action_net.weight.mul_(30.0)
    env = amd.GpuWaypointEnv(n, vehicle=vehicle, seed=4, max_episode_steps=60)                 # AUTO step kernel: the rollout brings its own env code
    ref = amd.GpuWaypointEnv(n, vehicle=vehicle, seed=4, max_episode_steps=60, kernel="team")  # the lane-quad step kernel: the same arithmetic
    assert "step_kernel_quad" in ref.kernel_name
    o0 = env.reset().clone(); ref.reset()
    dev = env.device
    b = _quad_rollout_buffers(T, n, dev)
    info = torch.zeros(T, n, dtype=torch.int32, device=dev); tobs = torch.full((T, n, 20), float("nan"), device=dev)
    env.rollout_policy(pol.flat_param, T, seed=77, draw0=5, info_bits=info, terminal_obs=tobs, **b)
    torch.cuda.synchronize()
    assert torch.equal(b["obs"][0], o0)
    lo, hi = pol.action_low, pol.action_high
    for t in range(T):
        o, r, d, i = ref.step(torch.max(torch.min(b["actions"][t], hi), lo))
        assert torch.equal(o, b["obs"][t + 1]) and torch.equal(r, b["rewards"][t]) and torch.equal(d, b["dones"][t]) and torch.equal(i, info[t]), t
        dn = d.bool()
        if bool(dn.any()):
            assert torch.equal(ref.terminal_obs[dn], tobs[t][dn])
    f1, i1 = env.get_state(); f2, i2 = ref.get_state()
    assert torch.equal(f1, f2) and torch.equal(i1, i2) and env.stats()["episodes"] == ref.stats()["episodes"] == int(b["dones"].sum()) > n // 2
    assert bool(torch.isnan(tobs[~b["dones"].bool()]).all())
    with torch.no_grad():
        flat = b["obs"][:T].reshape(T * n, 20)
        mean32 = pol.action_net(pol.mlp_extractor.policy_net(flat)); v32 = pol.value_net(pol.mlp_extractor.value_net(flat)).reshape(-1)
    std = torch.exp(pol.log_std.detach())
    assert float((b["values"].reshape(-1) - v32).abs().max()) < 3e-2 * max(1.0, float(v32.abs().max()))
    z = (b["actions"].reshape(T * n, 4) - mean32) / std
    assert abs(float(z.mean())) < 0.03 and abs(float(z.var()) - 1.0) < 0.04 and float(z.abs().max()) < 6.5
    assert float((torch.corrcoef(z[:50000].T) - torch.eye(4, device=dev)).abs().max()) < 0.04
    lp32 = (-0.5 * z * z - pol.log_std.detach() - 0.9189385332).sum(1)
    assert float((b["logp"].reshape(-1) - lp32).abs().max()) < 0.5 and float((b["logp"].reshape(-1) - lp32).abs().mean()) < 0.05
    env2 = amd.GpuWaypointEnv(n, vehicle=vehicle, seed=4, max_episode_steps=60); env2.reset()
    b2 = _quad_rollout_buffers(T, n, dev)
    env2.rollout_policy(pol.flat_param, T, seed=77, draw0=5, **b2)
    assert torch.equal(b2["actions"], b["actions"]) and torch.equal(b2["obs"], b["obs"])
    env.close(); ref.close(); env2.close()


def test_reference_checkpoint_inside_the_fused_rollout():
    """The reference's best checkpoint (v2/README.md:55; runsim_scaledObs.py:15) INSIDE the one-launch rollout: (1) teacher-forced on every state
    of the four recorded policy episodes, the kernel's deterministic action (log_std -> -20: sample == bf16 mean) is within 3e-2 of the action
    the reference recorded there; (2) sampled closed loop with the checkpoint's own log_std it still succeeds in > 95 % of its episodes, with the
    return / length of the step-by-step fp32 path."""
    from oracle import oracle as O
    from tests.golden_util import load, fill_blob
    pol = fixture_policy("cuda").flatten_()
    ls = pol.log_std.data.clone()
    worst = 0.0; rows = 0; errs = []
    for name in ("policy_ep0", "policy_ep1", "policy_ep2", "policy_ep3"):
        d = load(name)
        T = d["actions"].shape[0]
        env = amd.GpuWaypointEnv(T, seed=1)
        env.reset()
        f, i = env.get_state()
        f = f.cpu().numpy().astype(np.float64); i = i.cpu().numpy()
        fill_blob(f, i, d)
        env.set_state(f.astype(np.float32), i)
        b = _quad_rollout_buffers(1, T, env.device)
        with torch.no_grad():
            pol.log_std.data.fill_(-20.0)
        env.rollout_policy(pol.flat_param, 1, seed=0, draw0=0, **b)
        torch.cuda.synchronize()
        np.testing.assert_allclose(b["obs"][0].cpu().numpy()[1:], d["obs"][:-1], rtol=0, atol=2e-6)     # the kernel saw the recorded observations
        err = (torch.max(torch.min(b["actions"][0], pol.action_high), pol.action_low).cpu() - torch.from_numpy(d["actions"])).abs()
        worst = max(worst, float(err.max())); rows += T
        errs.append(err.reshape(-1))
        env.close()
    errs = torch.cat(errs)
    print(f"bf16 policy inside the rollout vs the recorded fp32 actions: max {worst:.4f}, mean {float(errs.mean()):.5f}, 99.9 % {float(errs.quantile(0.999)):.4f}")
    assert rows > 2000 and worst < 3e-2, worst
    # (2) closed loop, deterministic as the reference flies it (runsim_scaledObs.py:24 predict(deterministic=True)): the bf16 policy inside the launch
    n, T = 2048, 100
    env = amd.GpuWaypointEnv(n, seed=3)
    env.reset(); env.stats(reset=True)
    b = _quad_rollout_buffers(T, n, env.device)
    for k in range(17):
        env.rollout_policy(pol.flat_param, T, seed=9, draw0=k * T, **b)
    s = env.stats()
    assert s["episodes"] > 3000 and s["success"] / s["episodes"] > 0.95, s
    assert 12000 < s["return_sum"] / s["episodes"] < 26000 and 500 < s["length_sum"] / s["episodes"] < 900
    env.close()
    # (3) sampled with the checkpoint's own log_std (what a PPO rollout does): exploration noise costs successes -- as on the step-by-step fp32 path
    with torch.no_grad():
        pol.log_std.data.copy_(ls)
    rates = []
    for fused in (True, False):
        env = amd.GpuWaypointEnv(n, seed=3)
        obs = env.reset(); env.stats(reset=True)
        if fused:
            for k in range(17):
                env.rollout_policy(pol.flat_param, T, seed=9, draw0=k * T, **b)
        else:
            act = torch.zeros(n, 4, device=env.device); clipped = torch.zeros_like(act); lp = torch.zeros(n, device=env.device)
            for k in range(1700):
                mean, _ = pol.actor_critic(obs)
                gaussian_act(mean, pol.log_std.data, pol.action_low, pol.action_high, act, clipped, lp, 9, k, 0)
                obs = env.step(clipped)[0]
        st = env.stats()
        rates.append(st["success"] / st["episodes"])
        env.close()
    assert abs(rates[0] - rates[1]) < 0.03 and min(rates) > 0.8, rates


def test_ppo_fused_rollout_on_the_reference_vehicle():
    """PPO(quad_env, fused_rollout=True): the buffer the one-launch rollout fills is consistent (GAE of its own rewards / values / dones), the
    stored log-probs and values are those of the fp32 policy the update differentiates (ratio == 1 in the first epoch, as in SB3), an update runs."""
    from oracle import oracle as O
    env = amd.GpuWaypointEnv(512, seed=11, max_episode_steps=40)
    algo = PPO(env, policy=fixture_policy(env.device), n_steps=96, batch_size=4096, n_epochs=2, seed=5, fused_rollout=True)
    with torch.no_grad():
        algo.policy.log_std.data.fill_(-1.5)
    b = algo.collect_rollouts()
    T, n = algo.n_steps, algo.n_steps * env.num_envs
    with torch.no_grad():
        values, logp, _ = algo.policy.evaluate_actions(b.obs[:T].reshape(n, -1), b.actions.reshape(n, -1))
    assert float((logp - b.logp.reshape(n)).abs().max()) < 2e-4
    assert float((values - b.values.reshape(n)).abs().max()) < 1e-3 * max(1.0, float(values.abs().max()))
    d = b.dones.bool()
    assert 0 < int(d.sum()) < n // 4 and int(d[40].sum()) > 400
    adv_ref, _ = O.gae_reference(b.rewards.cpu().numpy(), b.values.cpu().numpy(), b.dones.cpu().numpy(), b.last_values.cpu().numpy(), 0.995, 0.9)
    mag = gae_magnitude(b.rewards.cpu().numpy(), b.values.cpu().numpy(), b.dones.cpu().numpy(), b.last_values.cpu().numpy())
    assert (np.abs(b.advantages.cpu().numpy() - adv_ref) / (1.0 + mag)).max() < 1e-6
    rec = algo.train()
    assert all(math.isfinite(x) for x in rec.values()) and rec["clip_fraction"] < 0.5


def test_pid_warm_start_moves_the_actor_towards_the_baseline():
    """amd.clone_pid_policy (the warm start PPO needs on the hexacopter vehicles, profiles/r03/ppo_hexa_*): a short cloning run lowers the actor's
    MSE against PidWaypointPolicy's actions along a PID flight, sets log_std, leaves the critic untouched, works for the arm vehicle's 29-D / 7-D
    interface too."""
    def mse_along_pid_flight(env, pol):
        pid = amd.PidWaypointPolicy.for_env(env)
        obs = env.reset(); done = None; tot = 0.0
        with torch.no_grad():
            for _ in range(150):
                a = pid.predict(obs, done)
                tot += float(((pol.actor(obs) - a) ** 2).mean())
                obs, _, done, _ = env.step(a)
        return tot / 150

    for vehicle in ("hexa", "hexa_arm"):
        env = amd.GpuWaypointEnv(256, vehicle=vehicle, seed=2)
        pol = ActorCritic(env.obs_dim, env.act_dim).cuda()
        critic0 = [p.detach().clone() for p in pol.mlp_extractor.value_net.parameters()]
        before = mse_along_pid_flight(env, pol)
        amd.clone_pid_policy(env, pol, steps=300, epochs=200, dagger_rounds=1)
        after = mse_along_pid_flight(env, pol)
        assert after < 0.7 * before, (vehicle, before, after)      # (the PID saturates its moment actions: the cloned MEAN cannot follow the chatter; measured 0.57 -> 0.31)
        assert float(pol.log_std.data.max()) == -1.0
        assert all(torch.equal(a, b) for a, b in zip(critic0, pol.mlp_extractor.value_net.parameters()))
        env.close()
