"""GPU parity: the HIP path (through the C ABI, via BatchedPlacementEnv) against
(a) the golden episodes recorded from the reference and (b) the CPU oracle on
device-sampled action streams.  Bit-exact: uint8 cells == reference 0/1,
float64 features / reward / info identical bit patterns."""
import os

import numpy as np
import pytest
import torch

from golden_util import case_names, load_case, pad_component_grid
from pcbenv import EnvConfig, InstanceStream, env_seed, named_config, pack_instances
from pcbenv.batched_env import BatchedPlacementEnv
from pcbenv.config import KIND_PIN, KIND_SPATIAL, KIND_SQUARE

pytestmark = pytest.mark.gpu

BEAM_READY = True


def _host(obs):
    return {k: v.cpu().numpy() for k, v in obs.items()}


def _same_bits(a, b):
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


@pytest.mark.parametrize("name", case_names())
def test_golden_episodes_on_gpu(name):
    meta, cfg, eps = load_case(name)
    if cfg.kind in (KIND_PIN, KIND_SPATIAL) and cfg.reward_type != "centroid" and not BEAM_READY:
        pytest.skip("beam routes not built yet")
    B = len(eps)
    env = BatchedPlacementEnv(cfg, B, queue_depth=1)
    if cfg.kind != KIND_SQUARE:
        env.load_instances([e.instance for e in eps])
    obs = _host(env.reset())
    for i, e in enumerate(eps):
        for k, stack in e.obs.items():
            want = stack[0] if k != "component_grid" else pad_component_grid(stack[0], cfg.max_num_components)
            assert np.array_equal(obs[k][i].astype(np.float64), want), (name, i, k, "reset")
    T = max(len(e.actions) for e in eps)
    for t in range(T):
        acts = np.zeros((B, 3), np.int32)
        live = [t < len(e.actions) for e in eps]
        for i, e in enumerate(eps):
            if live[i]:
                acts[i] = e.actions[t]
        o, r, d, info = env.step(torch.from_numpy(acts))
        obs = _host(o)
        r = r.cpu().numpy(); d = d.cpu().numpy()
        inf = env.info_raw.cpu().numpy()
        for i, e in enumerate(eps):
            if not live[i]:
                continue
            tag = (name, e.seed, e.ep, t, tuple(acts[i]))
            for k, stack in e.obs.items():
                want = stack[t + 1] if k != "component_grid" else pad_component_grid(stack[t + 1], cfg.max_num_components)
                got = obs[k][i].astype(np.float64)
                assert np.array_equal(got, want), (tag, k, np.argwhere(got != want)[:4])
            assert np.float64(r[i]).tobytes() == np.float64(e.reward[t]).tobytes(), (tag, r[i], e.reward[t])
            assert bool(d[i]) == bool(e.done[t]), tag
            if cfg.kind in (KIND_PIN, KIND_SPATIAL):
                if np.isnan(e.info[t, 0]):
                    assert np.isnan(inf[i]).all(), tag
                else:
                    assert _same_bits(inf[i], e.info[t]), (tag, inf[i], e.info[t])
    env.close()


def _oracle_rollout(cfg, B, episodes, queue_depth=2, p_bad=0.02, incremental=False, auto_reset=False, fused=False,
                    threads=0, cpu_threads=1, stats=None, max_steps=400, num_slots=1, device_instances=False, options=None,
                    stagger=0, compact=False, toggle_helpers=False):
    """Device-sampled legal actions (plus a few corrupted ones); every observation, reward, done, info of every step
    must equal the CPU oracle's.  Covers reset_done() and the instance queue.  cpu_threads > 1: the oracle steps and
    the whole-batch tensor comparison run under OpenMP (full-size batches).  stats: filled with counts of the
    terminal kinds seen.  stagger = L: during the first L steps environment i is reset once more after step
    i % L, so that from then on 1 / L of the batch ends an episode in every launch (the loop a policy runs)."""
    from oracle import oracle as orc
    env = BatchedPlacementEnv(cfg, B, queue_depth=queue_depth, run_seed=3, incremental_obs=incremental,
                              auto_reset=auto_reset, threads_per_env=threads, num_slots=num_slots, options=options, compact_features=compact)
    if device_instances:  # fresh instances from the on-device generator; the oracle takes the host generator's records
        from pcbenv.instances import NativeInstanceStreams
        env.enable_device_instances()
        host = NativeInstanceStreams(cfg, [env_seed(3, i) for i in range(B)])
        fresh = [host.next_packed() for _ in range(max_steps + 2)]  # an environment resets at most once per step
        packed = None
    else:
        packed = env.generate_instances(verify=4) if cfg.kind != KIND_SQUARE else None
    ob = orc.OracleBatch(cfg, B)
    cursor = np.zeros(B, np.int64)

    def oracle_reset(mask):
        if cfg.kind == KIND_SQUARE:
            for i in np.flatnonzero(mask):
                ob.env(i).reset()
            return
        if device_instances:
            rec = np.stack([fresh[min(cursor[i], len(fresh) - 1)][i] for i in range(B)])
            ob.reset_packed(rec, mask.astype(np.uint8), cpu_threads)
            cursor[mask.astype(bool)] += 1
            return
        rec = np.where((cursor % queue_depth == 0)[:, None], packed[0], packed[1 % queue_depth]) if queue_depth <= 2 else \
            np.stack([packed[cursor[i] % queue_depth][i] for i in range(B)])
        ob.reset_packed(rec, mask.astype(np.uint8), cpu_threads)
        cursor[mask.astype(bool)] += 1

    env.reset()
    oracle_reset(np.ones(B, np.uint8))
    if stats is not None:
        stats.update(worst_case_terminals=0, routed_terminals=0, env_steps=0)
    rng = np.random.RandomState(5)
    steps = 0
    done_eps = 0
    t = 0
    keys = list(env.obs.keys())
    while done_eps < episodes * B and t < max_steps:
        if toggle_helpers and t % 3 == 0:  # the terminal list's capacity changes under way (lists laid out for the old one are dropped)
            env.set_option("terminal_teams", (0, 16, 64, 48)[(t // 3) % 4])
        if num_slots > 1:  # trajectory layout: step t lands in slot (t + 1) % num_slots, slot 0 took the reset
            env.select_slot(t + 1)
        if fused:
            sampled = env.sample_actions(t).cpu().numpy()  # must equal what the fused launch draws
            o, r, d, _, a_dev = env.rollout_step(t)
            a = a_dev.cpu().numpy()
            assert np.array_equal(a, sampled), t
        else:
            a = env.sample_actions(t).cpu().numpy()
            bad = rng.rand(B) < p_bad
            a[bad] = rng.randint(-1, 70, size=(int(bad.sum()), 3))
            o, r, d, _ = env.step(torch.from_numpy(a))
        rr, dd, ii = ob.step(a, cpu_threads)
        if stats is not None and cfg.kind in (KIND_PIN, KIND_SPATIAL):
            worst = dd.astype(bool) & (ii[:, 0] == cfg.max_wirelength) & (ii[:, 1] == cfg.max_num_intersections)
            stats["worst_case_terminals"] += int(worst.sum())
            stats["routed_terminals"] += int(dd.sum() - worst.sum())
        if auto_reset:  # observations already show the next episode of the environments that finished
            oracle_reset(dd)
        obs = _host(env.obs_f64() if compact else o)  # compact feature tensors: compared after expansion to the reference's float64
        r = r.cpu().numpy(); d = d.cpu().numpy(); inf = env.info_raw.cpu().numpy()
        assert np.array_equal(d, dd), (t, np.flatnonzero(d != dd)[:5])
        assert _same_bits(r, rr), (t, np.flatnonzero(r != rr)[:5], r[r != rr][:3], rr[r != rr][:3])
        if cfg.kind in (KIND_PIN, KIND_SPATIAL):
            has = ~np.isnan(inf[:, 0])
            assert _same_bits(inf[has], ii[has]), t
        if cpu_threads > 1:  # whole batch, every tensor, compared inside the oracle library under OpenMP
            for k in keys:
                bad = ob.first_mismatch(k, obs[k], cpu_threads)
                assert bad < 0, (t, bad, k)
        else:
            for i in range(B):
                want = ob.env(i).obs()
                for k in keys:
                    assert np.array_equal(obs[k][i].astype(np.float64), want[k]), (t, i, k)
        done_eps += int(d.sum())
        if not auto_reset:
            env.reset_done()
            oracle_reset(d)
        if t < stagger:  # spread the episode phases: one more reset for a 1 / stagger slice of the batch
            m = (np.arange(B) % stagger == t).astype(np.uint8)
            env.reset(torch.from_numpy(m))
            oracle_reset(m)
        steps += B
        t += 1
    if device_instances:
        assert env.device_instance_errors() == 0 and int(cursor.max()) < len(fresh)
    env.close()
    if stats is not None:
        stats["env_steps"] = steps
    return steps


def test_c1_square_vs_oracle():
    _oracle_rollout(named_config("c1"), 8, episodes=3)


def test_c2_rect_vs_oracle():
    _oracle_rollout(named_config("c2"), 32, episodes=2)


def test_c3_pin_vs_oracle():
    _oracle_rollout(named_config("c3"), 32, episodes=2)


def test_c4_spatial_vs_oracle():
    _oracle_rollout(named_config("c4"), 16, episodes=2)


def test_c5_spatial_128_vs_oracle():
    _oracle_rollout(named_config("c5"), 4, episodes=1, queue_depth=1, p_bad=0.0)


def test_small_odd_grids_vs_oracle():
    _oracle_rollout(EnvConfig.spatial(10, 10, 3, 4, 2, 4, 2, 4, 6, 1, 2, 4, 5, 2, "centroid", 2, 0.5), 24, episodes=3, p_bad=0.05)
    _oracle_rollout(EnvConfig.pin(12, 12, 5, 5, 2, 5, 2, 5, 8, 6, 3, 5, 7, 2, "centroid", 2, 0.25), 24, episodes=3, p_bad=0.05)
    _oracle_rollout(EnvConfig.rect(6, 7, 2, 4, 2, 4, 4, 2), 16, episodes=3, p_bad=0.05)
    _oracle_rollout(EnvConfig.square(11, 10, 2), 8, episodes=2, p_bad=0.05)


def test_beam_and_both_rewards_vs_oracle():
    for rt, k in (("beam", 2), ("both", 2), ("both", 3), ("beam", 4)):
        _oracle_rollout(EnvConfig.pin(64, 64, 9, 9, 2, 6, 2, 6, 16, 16, 8, 8, 6, 6, rt, k, 0.5), 48, episodes=2, p_bad=0.0)
    _oracle_rollout(EnvConfig.spatial(128, 128, 9, 9, 2, 8, 2, 8, 32, 32, 16, 16, 8, 8, "both", 3, 0.5), 6, episodes=1, queue_depth=1, p_bad=0.0)
    _oracle_rollout(EnvConfig.spatial(24, 24, 5, 5, 2, 4, 2, 4, 12, 6, 2, 3, 16, 9, "both", 4, 0.5), 32, episodes=2, p_bad=0.02)


@pytest.mark.parametrize("name", ["c2", "c3", "c4"])
def test_incremental_obs_mode(name):
    _oracle_rollout(named_config(name), 24, episodes=2, incremental=True)


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c4"])
def test_auto_reset_and_fused_sampling(name):
    _oracle_rollout(named_config(name), 24, episodes=3, auto_reset=True)
    _oracle_rollout(named_config(name), 24, episodes=3, auto_reset=True, fused=True, incremental=True)
    _oracle_rollout(named_config(name), 24, episodes=2, fused=True)


def test_auto_reset_with_beam_routes():
    _oracle_rollout(named_config("c4", "both"), 16, episodes=3, auto_reset=True, fused=True)
    _oracle_rollout(EnvConfig.pin(12, 12, 5, 5, 2, 5, 2, 5, 8, 6, 3, 5, 7, 2, "beam", 2, 0.25), 24, episodes=3, p_bad=0.05, auto_reset=True)


@pytest.mark.parametrize("name,B,teams,fused", [("c3", 1024, None, False), ("c4", 512, None, False), ("c3", 1024, 3, True), ("c4", 256, 5, False),
                                                 ("c2", 1024, None, True), ("c3_both", 512, None, False), ("mid_beam", 384, 7, False),
                                                 ("c3", 1024, 0, False), ("c5", 256, None, False), ("c3_256threads", 256, None, True)])
def test_terminal_teams_with_staggered_episodes(name, B, teams, fused):
    """One launch per step, episode phases spread over the batch (1 / L of it terminal in every launch): the
    environments on the terminal list run on four-wavefront teams of k_step_mixed, the others on one wavefront each --
    every tensor of every environment at every step vs the oracle, with corrupted actions (terminal at once, off the
    list), lists longer than the teams' capacity (teams = 3 / 5 / 7: the rest falls back to its own wavefront) and
    the plain kernel (teams = 0)."""
    cfg = {"c3_both": lambda: named_config("c3", "both"), "c3_256threads": lambda: named_config("c3"),
           "mid_beam": lambda: EnvConfig.pin(12, 12, 5, 5, 2, 5, 2, 5, 8, 6, 3, 5, 7, 2, "beam", 2, 0.25)}.get(name, lambda: named_config(name))()
    L = cfg.max_num_components
    opts = None if teams is None else {"terminal_teams": teams}
    threads = 256 if name == "c3_256threads" else 0  # four-wavefront teams (c5's default): their helpers are four wavefronts too
    stats = {}
    _oracle_rollout(cfg, B, episodes=3 if L <= 16 else 2, queue_depth=3, p_bad=0.0 if fused else 0.01, auto_reset=True, fused=fused, cpu_threads=16,
                    stats=stats, max_steps=4 * L if L <= 16 else 3 * L, options=opts, stagger=L, threads=threads)
    if cfg.kind in (KIND_PIN, KIND_SPATIAL):
        assert stats["routed_terminals"] >= B
    # the explicit loop (step, then reset the finished ones): the terminal teams run reward-only terminal transitions
    _oracle_rollout(cfg, min(B, 256), episodes=2, queue_depth=2, p_bad=0.01, auto_reset=False, cpu_threads=16, max_steps=3 * L, options=opts, stagger=L)
    # trajectory layout
    _oracle_rollout(cfg, min(B, 128), episodes=2, queue_depth=2, p_bad=0.01, auto_reset=True, cpu_threads=16, max_steps=3 * L, options=opts, stagger=L,
                    num_slots=5)


def test_terminal_list_capacity_changes_under_way_and_option_errors():
    """pcbenv_set_option(PCBENV_OPT_TERMINAL_TEAMS) between steps -- off, small, larger -- with staggered episodes: results
    unchanged (the list is a scheduling hint); and what the option call refuses."""
    from pcbenv import _lib
    cfg = named_config("c3")
    _oracle_rollout(cfg, 512, episodes=3, queue_depth=3, p_bad=0.01, auto_reset=True, cpu_threads=16, max_steps=4 * cfg.max_num_components,
                    stagger=cfg.max_num_components, toggle_helpers=True)
    env = BatchedPlacementEnv(cfg, 8, queue_depth=1)
    for name, value in (("terminal_teams", -1), ("terminal_teams", 5000), ("stream_threshold_bytes", -5), ("gen_grid", 0), ("gen_lanes", 48)):
        with pytest.raises(ValueError):
            env.set_option(name, value)
    assert env._L.pcbenv_set_option(env._h, 99, 1) == _lib.PCBENV_EINVAL
    env.close()
    rect = BatchedPlacementEnv(named_config("c2"), 8, queue_depth=1)  # no routing reward to share
    with pytest.raises(ValueError):
        rect.set_option("terminal_teams", 16)
    rect.close()


@pytest.mark.parametrize("threads", [64, 256])
def test_threads_per_env_variants(threads):
    _oracle_rollout(named_config("c3"), 16, episodes=2, threads=threads)
    _oracle_rollout(named_config("c4", "both"), 16, episodes=2, threads=threads, auto_reset=True, fused=True)
    _oracle_rollout(named_config("c5"), 4, episodes=1, queue_depth=1, p_bad=0.0, threads=threads, incremental=True)
    _oracle_rollout(EnvConfig.spatial(10, 10, 3, 4, 2, 4, 2, 4, 6, 1, 2, 4, 5, 2, "beam", 2, 0.5), 16, episodes=2, threads=threads)


def test_state_dict_roundtrip_and_mask_bits():
    cfg = named_config("c4")
    env = BatchedPlacementEnv(cfg, 12, queue_depth=2, run_seed=9)
    env.generate_instances()
    env.reset()
    for t in range(5):
        env.step(env.sample_actions(t))
    snap = env.state_dict()
    bits = env.mask_bits().cpu().numpy().view(np.uint64)
    m = env.obs["action_mask"].cpu().numpy()
    for b in range(12):  # bit y of word [b, o, x, 0] == action_mask[b, o, x, y]
        for o in (0, 1):
            unpacked = ((bits[b, o, :, 0][:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.uint8)
            assert np.array_equal(unpacked, m[b, o])
    trace = []
    for t in range(5, 12):
        o, r, d, _ = env.step(env.sample_actions(t))
        trace.append((r.cpu().numpy().copy(), d.cpu().numpy().copy(), o["pin_grid"].cpu().numpy().copy()))
    env.load_state_dict(snap)
    for t in range(5, 12):
        o, r, d, _ = env.step(env.sample_actions(t))
        r0, d0, g0 = trace[t - 5]
        assert np.array_equal(r.cpu().numpy().view(np.uint64), r0.view(np.uint64)) and np.array_equal(d.cpu().numpy(), d0)
        assert np.array_equal(o["pin_grid"].cpu().numpy(), g0)
    env.close()


@pytest.mark.parametrize("name,B", [("c3", 64), ("c4", 32), ("c2", 64)])
def test_state_dict_roundtrip_with_the_device_generator(name, B):
    """Checkpoint / resume with a fresh on-device instance at every reset (the reference's reset() semantics): the blob
    carries the generator's two MT19937 streams per environment, its counters and the queued records, so the resumed run
    -- in the same handle, and in a fresh one -- replays more than three episodes bit for bit, observations included,
    and a second snapshot taken after the replay equals the one the original run would take."""
    cfg = named_config(name)
    L = cfg.max_num_components
    T = 3 * L + 5

    def run(env, t0):
        out = []
        for t in range(t0, t0 + T):
            o, r, d, _, a = env.rollout_step(t)
            out.append({"r": r.cpu().numpy().copy(), "d": d.cpu().numpy().copy(), "a": a.cpu().numpy().copy(),
                        **{k: v.cpu().numpy().copy() for k, v in o.items()}})
        return out

    env = BatchedPlacementEnv(cfg, B, queue_depth=6, run_seed=21, auto_reset=True)
    env.enable_device_instances()
    env.reset()
    for t in range(L + 3):
        env.rollout_step(t)
    snap = env.state_dict()
    assert snap["device_instances"] and snap["state"].size + snap["generator"].size == env._L.pcbenv_state_bytes(env._h) and snap["generator"].size > 0
    want = run(env, L + 3)
    after = env.state_dict()
    env.load_state_dict(snap)
    other = BatchedPlacementEnv(cfg, B, queue_depth=6, run_seed=99, auto_reset=True)  # other seeds: everything must come from the blob
    with pytest.raises(ValueError):
        other.load_state_dict(snap)  # the generator has to be enabled first
    other.enable_device_instances()
    other.reset()
    other.load_state_dict(snap)
    other.run_seed = env.run_seed  # (the key of the action sampler; the instance streams were seeded from 99 above)
    for e in (env, other):
        got = run(e, L + 3)
        for t, (g, w) in enumerate(zip(got, want)):
            for k in w:
                a, b = (g[k].view(np.uint64), w[k].view(np.uint64)) if g[k].dtype == np.float64 else (g[k], w[k])
                assert np.array_equal(a, b), (t, k)
        again = e.state_dict()
        assert np.array_equal(again["state"], after["state"]) and np.array_equal(again["generator"], after["generator"])
        assert e.device_instance_errors() == 0
        e.close()


def test_refused_host_load_leaves_the_generator_queue_alone():
    """Once the on-device generator owns the queue, pcbenv_load_instances is refused BEFORE anything is copied (the
    refusal used to come after the copy into the slot)."""
    from pcbenv import _lib
    cfg = named_config("c3")
    B = 32
    env = BatchedPlacementEnv(cfg, B, queue_depth=4, run_seed=5, auto_reset=True)
    packed = env.generate_instances()      # host records in every slot
    env.enable_device_instances()          # ... replaced by the generator's
    before = [env.queued_instances(s) for s in range(4)]
    assert not np.array_equal(before[0], packed[0][::-1])
    junk = np.ascontiguousarray(packed[0][::-1])
    for call in (lambda: env.load_packed(junk, 0), lambda: env.generate_instances(), lambda: env.refill_slot(1)):
        with pytest.raises(RuntimeError):
            call()
    rc = env._L.pcbenv_load_instances(env._h, None, B, 0, junk.ctypes.data, env._stream())  # straight through the C ABI
    assert rc == _lib.PCBENV_ESTATE
    for s in range(4):
        assert np.array_equal(env.queued_instances(s), before[s]), s
    env.close()


def test_rollout_driver_matches_oracle_returns():
    """Trajectory buffers on device; per-episode returns of the uniform random policy == the oracle's on the
    recorded action stream (counterpart of the reference's simulate() loop)."""
    from oracle import oracle as orc
    from pcbenv import rollout
    cfg = named_config("c3")
    B, T, Q = 16, 40, 3
    env = BatchedPlacementEnv(cfg, B, queue_depth=Q, run_seed=4, auto_reset=True)
    packed = env.generate_instances(native=False)
    env.reset()
    traj = rollout.collect(env, T, store_obs=("placement_mask",))
    ob = orc.OracleBatch(cfg, B)
    cursor = np.zeros(B, np.int64)

    def oreset(mask):
        ob.reset_packed(np.stack([packed[cursor[i] % Q][i] for i in range(B)]), mask.astype(np.uint8))
        cursor[mask.astype(bool)] += 1
    oreset(np.ones(B, np.uint8))
    acts = traj.actions.cpu().numpy()
    for t in range(T):
        r, d, _ = ob.step(acts[t])
        assert np.array_equal(r.view(np.uint64), traj.rewards[t].cpu().numpy().view(np.uint64))
        assert np.array_equal(d, traj.dones[t].cpu().numpy())
        oreset(d)
    rets = traj.episode_returns()
    assert all(len(e) == 2 for e in rets) and all(x < 0 for e in rets for x in e)  # 40 steps = 2 full episodes of 16
    # a policy callable: masked categorical over flat logits never picks an illegal action
    env2 = BatchedPlacementEnv(cfg, B, queue_depth=1, run_seed=4)
    env2.generate_instances(); env2.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    pol = lambda obs: rollout.sample_masked_categorical(torch.zeros((B, 4 * 64 * 64), device="cuda"), obs["action_mask"].reshape(B, -1), g)
    tr2 = rollout.collect(env2, 16, policy=pol)
    assert int(tr2.dones[:15].sum()) == 0 and int(tr2.dones[15].sum()) == B
    env.close(); env2.close()


def test_mask_marginals_for_factorised_policies():
    """reduce_max over (H, W) and over W of action_mask (factorized_action_distributions.py:358, :401)."""
    for name, threads in (("c2", 0), ("c4", 0), ("c5", 256), ("c1", 0)):
        cfg = named_config(name)
        B = 6
        env = BatchedPlacementEnv(cfg, B, queue_depth=1, run_seed=2, mask_marginals=True, threads_per_env=threads)
        env.generate_instances()
        env.reset()
        for t in range(10):
            m = env.obs["action_mask"].reshape(B, cfg.num_orientations, cfg.height, cfg.width)
            assert torch.equal(env.mask_marginals["orientation"], m.amax(dim=(2, 3)))
            assert torch.equal(env.mask_marginals["rows"], m.amax(dim=3))
            env.step(env.sample_actions(t))
        env.close()


def test_ppo_loop_runs_and_uses_only_legal_actions():
    """Two PPO iterations close the loop on the GPU: finite losses, and the masked policy never ends an episode
    with an invalid action (every episode of the 2x2-component config has exactly 5 steps)."""
    from pcbenv.policy import SpatialPolicy
    from pcbenv.ppo import PPOConfig, PPOTrainer
    torch.manual_seed(0)
    cfg = EnvConfig.spatial(10, 10, 9, 9, 2, 2, 2, 2, 5, 5, 3, 3, 6, 6, "centroid", 2, 0.75)
    env = BatchedPlacementEnv(cfg, 64, queue_depth=4, auto_reset=True)
    env.generate_instances()
    env.reset()
    tr = PPOTrainer(env, SpatialPolicy(cfg).to(env.device), PPOConfig(rollout_steps=10, epochs=1, minibatches=2))
    batch = tr.collect()
    assert float(batch["obs"]["action_mask"].float().sum()) > 0
    stats = tr.update(batch)
    assert all(np.isfinite(v) for v in stats.values())
    tr.update(tr.collect())
    assert len(tr.returns) == 2 and all(-11.0 < r < 0.0 for r in tr.returns)  # worst case (invalid action) is -10.6
    env.close()


def test_rollout_steps_equals_single_steps():
    cfg = named_config("c2")
    envs = [BatchedPlacementEnv(cfg, 32, queue_depth=2, run_seed=5, auto_reset=True) for _ in range(2)]
    for e in envs:
        e.generate_instances(); e.reset()
    acts = envs[0].rollout_steps(0, 20).cpu().numpy()
    for t in range(20):
        _, _, _, _, a = envs[1].rollout_step(t)
        assert np.array_equal(a.cpu().numpy(), acts[t])
    for k in envs[0].obs:
        assert torch.equal(envs[0].obs[k], envs[1].obs[k])
    assert torch.equal(envs[0].reward, envs[1].reward) and torch.equal(envs[0].done, envs[1].done)
    for e in envs:
        e.close()


def test_ragged_and_maximum_shapes_vs_oracle():
    """Edge shapes: non-square grids whose width is not a multiple of 16 / spans two 64-bit words, a 1-component
    instance range, and the largest configuration the HIP path accepts (128x128, 64 components, 256 pins, 16 nets
    of 16 pins, beam width 4)."""
    _oracle_rollout(EnvConfig.spatial(7, 100, 5, 5, 2, 7, 2, 7, 10, 1, 1, 4, 6, 2, "both", 2, 0.5), 12, episodes=2, p_bad=0.05)
    _oracle_rollout(EnvConfig.pin(100, 9, 5, 5, 2, 6, 2, 6, 12, 3, 2, 4, 6, 2, "centroid", 2, 0.5), 12, episodes=2, p_bad=0.05)
    _oracle_rollout(EnvConfig.rect(33, 65, 1, 9, 1, 9, 40, 5), 8, episodes=2, p_bad=0.02)
    _oracle_rollout(EnvConfig.square(3, 130 - 2, 3), 4, episodes=1, p_bad=0.0)
    big = EnvConfig.spatial(128, 128, 9, 9, 2, 8, 2, 8, 64, 40, 8, 16, 16, 4, "both", 4, 0.5)
    assert big.max_total_pins == 256
    _oracle_rollout(big, 3, episodes=1, queue_depth=1, p_bad=0.0)
    _oracle_rollout(big, 3, episodes=1, queue_depth=1, p_bad=0.0, threads=64, auto_reset=True, fused=True)
    # a grid completely filled by one component, and 1x1 components
    _oracle_rollout(EnvConfig.rect(4, 4, 4, 4, 4, 4, 2, 2), 4, episodes=2, p_bad=0.0)
    _oracle_rollout(EnvConfig.rect(6, 6, 1, 1, 1, 1, 3, 1), 4, episodes=2, p_bad=0.1)
    # components as large as the largest grid (128-bit row folds with shifts >= 64)
    _oracle_rollout(EnvConfig.rect(128, 128, 1, 128, 1, 128, 6, 1), 6, episodes=3, p_bad=0.0)
    _oracle_rollout(EnvConfig.rect(128, 128, 40, 100, 30, 90, 6, 2), 6, episodes=3, p_bad=0.02, threads=256)


@pytest.mark.parametrize("name,B", [("c3", 4096), ("c4", 4096), ("c5", 8192)])
def test_full_size_batches(name, B):
    """BASELINE.json's full batch sizes.  (1) Two independent runs in different modes -- explicit three-call loop
    with full refresh vs the fused one-launch loop with incremental rows -- must produce identical tensors for
    every environment at every step (mode-independence at full size).  (2) Size-independent invariants: occupied
    cells == sum of placed component areas, mask planes 2/3 mirror 0/1, a legal cell is an empty cell, every
    episode of these configs ends exactly when the last component is placed, pin_grid == one-hot of the pins.
    (3) A random sample of 96 environments is replayed through the CPU oracle on the recorded actions."""
    from oracle import oracle as orc
    cfg = named_config(name)
    T = cfg.max_num_components + 3
    a = BatchedPlacementEnv(cfg, B, queue_depth=2, run_seed=11)
    b = BatchedPlacementEnv(cfg, B, queue_depth=2, run_seed=11, auto_reset=True, incremental_obs=True)
    packed = a.generate_instances(verify=8)
    b.generate_instances()
    a.reset(); b.reset()
    acts, rews, dones = [], [], []
    area = None
    for t in range(T):
        act = a.sample_actions(t)
        a.step(act)
        _, _, _, _, act_b = b.rollout_step(t)
        assert torch.equal(act, act_b), t
        bad = ((a.reward != b.reward) | (a.done != b.done)).nonzero().flatten()
        assert len(bad) == 0, (t, len(bad), bad[:8].tolist(), a.reward[bad[:4]].tolist(), b.reward[bad[:4]].tolist(), a.done[bad[:4]].tolist(), b.done[bad[:4]].tolist())
        feat = a.obs["all_components_feature"]
        placed = (feat[:, :, 2] >= 0).double()
        area = (feat[:, :, 0] * feat[:, :, 1] * placed).sum(dim=1)
        assert torch.equal(a.obs["grid"].sum(dim=(1, 2)).double(), area), t
        m = a.obs["action_mask"]
        assert torch.equal(m[:, 0], m[:, 2]) and torch.equal(m[:, 1], m[:, 3]), t
        assert int((m[:, 0] & a.obs["grid"]).sum()) == 0 and int((m[:, 1] & a.obs["grid"]).sum()) == 0, t
        if "pin_grid" in a.obs:
            pg = a.obs["pin_grid"]
            assert int(pg.sum(dim=3).max()) <= 1 and torch.equal(pg.sum(dim=3), a.obs["grid"]), t
        steps_in_ep = t % cfg.max_num_components
        assert bool((a.done == (1 if steps_in_ep == cfg.max_num_components - 1 else 0)).all()), t
        acts.append(act.cpu().numpy()); rews.append(a.reward.cpu().numpy()); dones.append(a.done.cpu().numpy())
        a.reset_done()
        for k in a.obs:  # after the explicit reset both runs show the same observations again
            assert torch.equal(a.obs[k], b.obs[k]), (t, k)
    # oracle replay of a sample
    rng = np.random.RandomState(0)
    idx = np.sort(rng.choice(B, 96, replace=False))
    ob = orc.OracleBatch(cfg, len(idx))
    cursor = 0
    ob.reset_packed(packed[0][idx])
    for t in range(T):
        r, d, _ = ob.step(acts[t][idx], threads=4)
        assert np.array_equal(r.view(np.uint64), rews[t][idx].view(np.uint64)), t
        assert np.array_equal(d, dones[t][idx]), t
        if d.any():
            cursor += 1
            ob.reset_packed(packed[cursor % 2][idx], d.astype(np.uint8))
    a.close(); b.close()


@pytest.mark.parametrize("name,B,stream_mb", [
    ("c2", 1024, None), ("c3", 4096, None), ("c3", 4096, 0), ("c4", 4096, None), ("c4", 4096, 0),
    ("c5", 8192, None), ("c5", 2048, 1 << 20)])
def test_every_tensor_at_the_baseline_batches(name, B, stream_mb, monkeypatch):
    """BASELINE.json's batch sizes, one whole episode (plus the reset that follows it) in the bench's own mode (fused
    sampling + in-launch reset): EVERY observation tensor, reward, done and info of EVERY environment at every step
    against the CPU oracle.  stream_mb forces the other store flavour (0: streaming `sc1 nt` stores below the
    Infinity-Cache threshold; 1 << 20: write-through `sc1` stores for a c5 batch that would otherwise stream)."""
    cfg = named_config(name)
    stats = {}
    n = _oracle_rollout(cfg, B, episodes=1, queue_depth=2, p_bad=0.0, auto_reset=True, fused=True, cpu_threads=16,
                        stats=stats, max_steps=cfg.max_num_components + 2,
                        options=None if stream_mb is None else {"stream_threshold_bytes": stream_mb << 20})
    assert n >= B * cfg.max_num_components
    if cfg.kind in (KIND_PIN, KIND_SPATIAL):
        assert stats["routed_terminals"] >= B


@pytest.mark.parametrize("threads,incremental", [(256, True), (256, False), (64, True)])
def test_no_legal_cell_terminals_with_in_launch_reset(threads, incremental):
    """Crowded grids (12x12, eight components of up to 5x5): many episodes end because the NEXT component has no legal
    cell (Q8 ii), so the step's own grid / pin_grid / mask stores are not skipped and the in-launch reset rewrites the
    same bytes from other lanes and -- threads_per_env = 256 -- other wavefronts.  Every tensor against the oracle."""
    cfg = EnvConfig.spatial(12, 12, 5, 5, 2, 5, 2, 5, 8, 8, 3, 5, 7, 2, "centroid", 2, 0.5)
    stats = {}
    _oracle_rollout(cfg, 512, episodes=6, queue_depth=2, p_bad=0.0, incremental=incremental, auto_reset=True,
                    fused=True, threads=threads, cpu_threads=8, stats=stats, max_steps=60)
    assert stats["worst_case_terminals"] > 200 and stats["routed_terminals"] > 20, stats
    cfgp = EnvConfig.pin(12, 12, 5, 5, 2, 5, 2, 5, 8, 8, 3, 5, 7, 2, "both", 2, 0.5)
    stats = {}
    _oracle_rollout(cfgp, 256, episodes=4, queue_depth=2, p_bad=0.0, incremental=incremental, auto_reset=True,
                    fused=True, threads=threads, cpu_threads=8, stats=stats, max_steps=40)
    assert stats["worst_case_terminals"] > 50, stats


@pytest.mark.parametrize("name,mode", [("c2", "explicit"), ("c3", "explicit"), ("c4", "explicit"), ("c4", "fused"), ("c3", "fused"),
                                       ("small_spatial", "explicit"), ("small_spatial", "fused"), ("small_pin", "fused"), ("c1", "fused")])
def test_trajectory_slots_every_tensor(name, mode):
    """[num_slots, B, ...] layout (pcbenv_bind_buffers_slots): every step lands in its own slot and must write every
    tensor whole -- the float64 feature tensors that the in-place layout maintains row by row, component_grid
    (with the rotation of placed components undone, Q4), the unchanged observation after an invalid action."""
    cfg = {"small_spatial": EnvConfig.spatial(10, 10, 3, 4, 2, 4, 2, 4, 6, 1, 2, 4, 5, 2, "both", 2, 0.5),
           "small_pin": EnvConfig.pin(10, 10, 3, 4, 2, 4, 2, 4, 6, 1, 2, 4, 5, 2, "centroid", 2, 0.5)}.get(name) or named_config(name)
    fused = mode == "fused"
    _oracle_rollout(cfg, 96, episodes=3, queue_depth=2, p_bad=0.0 if fused else 0.03, auto_reset=fused, fused=fused,
                    num_slots=5, max_steps=60)
    if cfg.kind != KIND_SQUARE:  # the same with the compact feature tensors (int16 / int8 / uint8), expanded for the comparison
        _oracle_rollout(cfg, 96, episodes=3, queue_depth=2, p_bad=0.0 if fused else 0.03, auto_reset=fused, fused=fused,
                        num_slots=5, max_steps=60, compact=True)


@pytest.mark.parametrize("name,B,T,S,compact", [("c3", 512, 40, 41, False), ("c4", 256, 36, 37, False), ("c2", 512, 20, 7, False), ("c5", 64, 40, 41, False),
                                                ("small_spatial", 256, 30, 31, False), ("c1", 64, 12, 13, False),
                                                # BASELINE batches, one episode + the reset behind it, every tensor of every slot
                                                ("c3", 4096, 17, 18, False), ("c4", 4096, 17, 18, False), ("c2", 1024, 17, 18, False), ("c5", 512, 33, 34, False),
                                                # compact feature tensors (pcbenv_bind_compact_features), expanded for the comparison
                                                ("c3", 4096, 17, 18, True), ("c4", 4096, 17, 18, True), ("c2", 512, 20, 7, True), ("c5", 64, 40, 41, True),
                                                ("small_spatial", 256, 30, 31, True)])
def test_persistent_rollout_fills_the_trajectory(name, B, T, S, compact):
    """pcbenv_rollout_sampled = ONE launch for T steps with the state held in LDS: slot (1 + t) % S of every tensor, the
    recorded actions, rewards, dones and infos against the oracle stepping the same actions.  (S < T: the slots wrap.)"""
    from oracle import oracle as orc
    cfg = EnvConfig.spatial(10, 10, 3, 4, 2, 4, 2, 4, 6, 1, 2, 4, 5, 2, "both", 2, 0.5) if name == "small_spatial" else named_config(name)
    Q = 8
    env = BatchedPlacementEnv(cfg, B, queue_depth=Q, run_seed=11, auto_reset=True, num_slots=S, compact_features=compact)
    packed = env.generate_instances(verify=4) if cfg.kind != KIND_SQUARE else None
    ob = orc.OracleBatch(cfg, B)
    cursor = np.zeros(B, np.int64)

    def oracle_reset(mask):
        if cfg.kind == KIND_SQUARE:
            for i in np.flatnonzero(mask):
                ob.env(i).reset()
            return
        sel = cursor % Q
        rec = packed[0].copy()
        for q_ in range(1, Q):
            rec[sel == q_] = packed[q_][sel == q_]
        ob.reset_packed(rec, mask.astype(np.uint8), 16)
        cursor[mask.astype(bool)] += 1

    env.reset()                       # slot 0
    oracle_reset(np.ones(B, np.uint8))
    env.select_slot(1)
    acts = env.rollout_steps(0, T).cpu().numpy()
    if compact:
        from pcbenv.batched_env import FEATURE_KEYS, expand_compact_features
        assert all(env.traj[k].dtype in (torch.int16, torch.int8, torch.uint8) for k in env.traj if k in FEATURE_KEYS)
        wide = expand_compact_features(cfg, {k: v for k, v in env.traj.items() if k in FEATURE_KEYS})
        traj = {k: (wide[k] if k in wide else v).cpu().numpy() for k, v in env.traj.items()}
    else:
        traj = {k: v.cpu().numpy() for k, v in env.traj.items()}
    rew, done, info = env.traj_reward.cpu().numpy(), env.traj_done.cpu().numpy(), env.traj_info.cpu().numpy()
    single = BatchedPlacementEnv(cfg, B, queue_depth=Q, run_seed=11, auto_reset=True)  # the same draws, one launch per step
    if packed is not None:
        for s_, pk in enumerate(packed):
            single.load_packed(pk, slot=s_)
    single.reset()
    for t in range(T):
        a1 = single.rollout_step(t)[4].cpu().numpy()
        assert np.array_equal(a1, acts[t]), t
        rr, dd, ii = ob.step(acts[t], 16)
        oracle_reset(dd)
        if t + S < T:
            continue                  # this slot has been overwritten by a later step
        s_ = (1 + t) % S
        assert np.array_equal(done[s_], dd), t
        assert _same_bits(rew[s_], rr), t
        if cfg.kind in (KIND_PIN, KIND_SPATIAL):
            has = dd.astype(bool)
            assert _same_bits(info[s_][has], ii[has]) and np.isnan(info[s_][~has]).all(), t
        for k in traj:
            assert ob.first_mismatch(k, traj[k][s_], 16) < 0, (t, k)
    env.close(); single.close()


GEN_CASES = {
    "c2": lambda: named_config("c2"), "c3": lambda: named_config("c3"), "c4": lambda: named_config("c4"),
    "c5": lambda: named_config("c5"),
    # min < max everywhere: truncated multinomial (step 7), clamped component counts, net / pin clipping
    "small_spatial": lambda: EnvConfig.spatial(10, 10, 3, 4, 2, 4, 2, 4, 6, 1, 2, 4, 5, 2, "centroid", 2, 0.5),
    "small_pin": lambda: EnvConfig.pin(10, 10, 3, 4, 2, 4, 2, 4, 6, 1, 2, 4, 5, 2, "centroid", 2, 0.5),
    "mid_spatial": lambda: EnvConfig.spatial(12, 12, 5, 5, 2, 5, 2, 5, 8, 6, 3, 5, 7, 2, "centroid", 2, 0.25),
    "rect_6x6": lambda: EnvConfig.rect(6, 6, 2, 4, 2, 4, 4, 2),
}


@pytest.mark.parametrize("name,B,lanes", [("c2", 2048, 0), ("c3", 4096, 0), ("c4", 1024, 0), ("c5", 512, 0), ("small_spatial", 4096, 0),
                                          ("small_pin", 4096, 0), ("mid_spatial", 2048, 0), ("rect_6x6", 1024, 0),
                                          ("c3", 1024, 32), ("c3", 1024, 64), ("c5", 256, 64), ("small_spatial", 1024, 64)])
def test_device_instance_generator_equals_the_host_streams(name, B, lanes, monkeypatch):
    """k_gen_fill against csrc/instance_gen.cpp (itself pinned to the reference's generate_instances by
    test_instance_gen_native.py and the golden tables): the whole queue after enabling, and -- after rollouts that
    consume and refill it several times over -- every record an environment is about to take."""
    from pcbenv.instances import NativeInstanceStreams
    cfg = GEN_CASES[name]()
    Q = 8
    # lanes per environment of the generator kernel (default: the narrowest group the configuration allows)
    env = BatchedPlacementEnv(cfg, B, queue_depth=Q, run_seed=7, auto_reset=True, options={"gen_lanes": lanes} if lanes else None)
    env.enable_device_instances()
    host = NativeInstanceStreams(cfg, [env_seed(7, i) for i in range(B)])
    K = 80
    recs = [host.next_packed() for _ in range(K)]
    for s_ in range(Q):
        got = env.queued_instances(s_)
        bad = np.flatnonzero((got != recs[s_]).any(axis=1))
        assert len(bad) == 0, (name, s_, bad[:5])
    env.reset()
    for t in range(60):
        env.rollout_step(t)
    assert env.device_instance_errors() == 0
    sd = env.state_dict()["state"].reshape(B, -1)
    cursor = sd[:, 12:16].copy().view(np.uint32)[:, 0].astype(np.int64)   # EnvHdr::qcursor
    assert cursor.min() >= 2 and cursor.max() < K - Q, (cursor.min(), cursor.max())
    slots = [env.queued_instances(s_) for s_ in range(Q)]
    for i in range(0, B, 3):  # device_instance_errors() brought the queue up to date: records cursor .. cursor + Q - 1
        for ahead in range(Q):
            r = int(cursor[i]) + ahead
            assert np.array_equal(slots[r % Q][i], recs[r][i]), (name, i, r)
    env.close()


@pytest.mark.parametrize("name,B,mode", [("c3", 256, "fused"), ("c4", 128, "fused"), ("c2", 256, "explicit"), ("small_spatial", 256, "fused"),
                                         ("small_pin", 256, "explicit"), ("mid_spatial", 128, "fused")])
def test_fresh_instances_every_reset_vs_oracle(name, B, mode):
    """The reference's reset() semantics at full speed: every episode of every environment on a NEW instance from its
    stream (on-device generator, queue refilled on the side stream while the steps run); every tensor, reward, done
    and info against the oracle resetting from the host generator's records."""
    cfg = GEN_CASES[name]()
    fused = mode == "fused"
    _oracle_rollout(cfg, B, episodes=7, queue_depth=4, p_bad=0.0 if fused else 0.02, auto_reset=fused, fused=fused,
                    device_instances=True, max_steps=150)


def test_flat_actions_equal_tuple_actions():
    """PCBENV_ACTION_FLAT (utils/environment/env_wrappers.py:80-98, :184-199) decodes to the same transition as the
    tuple format, including out-of-range flat indices (invalid -> terminal)."""
    for name in ("c1", "c2", "c4"):
        cfg = named_config(name)
        B = 24
        envs = [BatchedPlacementEnv(cfg, B, queue_depth=2, run_seed=6) for _ in range(2)]
        for e in envs:
            e.generate_instances(); e.reset()
        O, H, W = cfg.num_orientations, cfg.height, cfg.width
        rng = np.random.RandomState(1)
        for t in range(2 * max(cfg.max_num_components, 4)):
            a = envs[0].sample_actions(t)
            flat = (a[:, 0] * H + a[:, 1]) * W + a[:, 2] if O > 1 else a[:, 1] * W + a[:, 2]
            flat = flat.clone()
            bad = torch.from_numpy(rng.rand(B) < 0.05).to(flat.device)
            a = a.clone()
            a[bad] = torch.tensor([0, H, 0], dtype=torch.int32, device=a.device)     # row out of range
            flat[bad] = O * H * W + 3                                                 # flat index out of range
            envs[0].step(a)
            envs[1].step(flat)
            for k in envs[0].obs:
                assert torch.equal(envs[0].obs[k], envs[1].obs[k]), (name, t, k)
            assert torch.equal(envs[0].reward, envs[1].reward) and torch.equal(envs[0].done, envs[1].done)
            envs[0].reset_done(); envs[1].reset_done()
        for e in envs:
            e.close()


def test_queue_refill_continues_the_instance_streams():
    """queue_depth = 1 with refill_slot between episodes: episode k of environment i plays the k-th instance of its
    reference stream (native generator), exactly like k successive reset() calls of the reference env."""
    from oracle import oracle as orc
    cfg = named_config("c3")
    B = 8
    env = BatchedPlacementEnv(cfg, B, queue_depth=1, run_seed=7)
    first = env.generate_instances()[0]
    streams = [InstanceStream(cfg, env_seed(7, i)) for i in range(B)]
    want0 = pack_instances(cfg, [s.next() for s in streams])
    assert np.array_equal(first, want0)
    env.reset()
    ob = orc.OracleBatch(cfg, B)
    ob.reset_packed(want0)
    for ep in range(3):
        for t in range(cfg.max_num_components):
            a = env.sample_actions(ep * 100 + t)
            _, r, d, _ = env.step(a)
            rr, dd, _ = ob.step(a.cpu().numpy())
            assert np.array_equal(r.cpu().numpy().view(np.uint64), rr.view(np.uint64)) and np.array_equal(d.cpu().numpy(), dd)
        assert bool(env.done.all())
        nxt = env.refill_slot(0)
        want = pack_instances(cfg, [s.next() for s in streams])
        assert np.array_equal(nxt, want), ep
        env.reset_done()
        ob.reset_packed(want)
        got = env.obs["all_components_feature"].cpu().numpy()
        for i in range(B):
            assert np.array_equal(got[i], ob.env(i).obs()["all_components_feature"])
    env.close()


def test_instance_feeder_keeps_streams_in_order():
    """Background generation + stream-ordered refills: over 7 episodes with a 2-deep queue every environment plays
    instance k of its reference stream in episode k (checked through the oracle on the same streams)."""
    from oracle import oracle as orc
    from pcbenv.feeder import InstanceFeeder
    cfg = named_config("c3")
    B, Q = 16, 2
    env = BatchedPlacementEnv(cfg, B, queue_depth=Q, run_seed=8, auto_reset=True)
    env.generate_instances()
    env.reset()
    assert env.queue_cursors() == (1, 1)
    streams = [InstanceStream(cfg, env_seed(8, i)) for i in range(B)]
    ob = orc.OracleBatch(cfg, B)
    ob.reset_packed(pack_instances(cfg, [s.next() for s in streams]))
    with InstanceFeeder(env, prefetch=3) as feeder:
        for ep in range(7):
            for t in range(cfg.max_num_components):
                _, r, d, _, a = env.rollout_step(ep * 64 + t)
                rr, dd, _ = ob.step(a.cpu().numpy())
                assert np.array_equal(r.cpu().numpy().view(np.uint64), rr.view(np.uint64)) and np.array_equal(d.cpu().numpy(), dd)
            assert bool(env.done.all())
            ob.reset_packed(pack_instances(cfg, [s.next() for s in streams]))   # what the reference would draw next
            got = env.obs["all_components_feature"].cpu().numpy()
            for i in range(B):
                assert np.array_equal(got[i], ob.env(i).obs()["all_components_feature"]), (ep, i)
            feeder.refill(block=True)
    assert env.queue_cursors() == (8, 8)
    env.close()


def _mix64(z):
    M = (1 << 64) - 1
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return z ^ (z >> 31)


def test_sampler_definition():
    """The uniform legal-action draw is a documented function of (seed, global env index, step index, mask bits):
    rnd = mix64(mix64(seed ^ GOLDEN*(env+1)) + step); pick = floor(hi32(rnd) * n / 2^32) over the n = reps*popcount
    legal actions in (orientation, row, column) order -- checked here against a host computation from mask_bits(),
    independently of how the kernel scans and selects."""
    M = (1 << 64) - 1
    for name, first in (("c1", 0), ("c2", 7), ("c4", 1000), ("c5", 3)):
        cfg = named_config(name)
        B = 12
        env = BatchedPlacementEnv(cfg, B, queue_depth=1, run_seed=5, first_env_index=first)
        env.generate_instances(); env.reset()
        H, W = cfg.height, cfg.width
        nplanes = 1 if cfg.kind == KIND_SQUARE else 2
        reps = 2 if cfg.kind in (KIND_PIN, KIND_SPATIAL) else 1
        for t in range(cfg.max_num_components + 2 if cfg.kind != KIND_SQUARE else 6):
            bits = env.mask_bits().cpu().numpy().view(np.uint64)[:, :nplanes]          # [B, planes, H, WW]
            a = env.sample_actions(t).cpu().numpy()
            for e in range(B):
                words = bits[e].reshape(-1)
                setbits = [(wi, b) for wi, w in enumerate(words) for b in range(64) if (int(w) >> b) & 1]
                total = len(setbits)
                if total == 0:
                    assert tuple(a[e]) == (0, 0, 0)
                    continue
                rnd = _mix64((_mix64((5 ^ ((0x9E3779B97F4A7C15 * (first + e + 1)) & M)) & M) + t) & M)
                pick = ((rnd >> 32) * (total * reps)) >> 32
                k, rep = pick % total, pick // total
                wi, b = setbits[k]
                WW = (W + 63) // 64
                plane, rw = divmod(wi, H * WW)
                x, wcol = divmod(rw, WW)
                want = (plane + 2 * rep, x, wcol * 64 + b)
                assert tuple(int(v) for v in a[e]) == want, (name, t, e, tuple(a[e]), want)
            env.step(torch.from_numpy(a))
            env.reset_done()
        env.close()


def test_step_is_graph_capturable_with_a_policy():
    """pcbenv_step enqueues kernels on the caller's stream and nothing else (no allocation, copy or synchronisation), so
    a policy forward + step can be captured once in a HIP graph (torch.cuda.CUDAGraph) and replayed: the replays
    must produce what the eager loop produces, across the in-launch resets."""
    cfg = named_config("c4")
    B = 256

    def make():
        env = BatchedPlacementEnv(cfg, B, queue_depth=2, auto_reset=True, run_seed=1)
        env.generate_instances(); env.reset()
        return env

    torch.manual_seed(0)
    pol = torch.nn.Conv2d(1, 4, 3, padding=1).cuda()
    HW = cfg.height * cfg.width

    def policy_step(env, out):
        with torch.no_grad():
            logits = pol(env.obs["grid"].float().unsqueeze(1)).flatten(1)        # [B, 4*H*W], (o, x, y) order
            flat = logits.masked_fill(~env.obs["action_mask"].flatten(1).bool(), -1e9).argmax(dim=1)
            out[:, 0] = (flat // HW).int(); out[:, 1] = ((flat % HW) // cfg.width).int(); out[:, 2] = (flat % cfg.width).int()
        env.step(out)

    a, w = make(), make()
    acts_a = torch.empty((B, 3), dtype=torch.int32, device="cuda")
    acts_b = torch.empty((B, 3), dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):  # warm-up off the default stream, as torch asks for before a capture
        for _ in range(3):
            policy_step(w, acts_b)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    w.close()
    b = make()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        policy_step(b, acts_b)
    for t in range(cfg.max_num_components + 4):
        policy_step(a, acts_a)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(acts_a, acts_b), t
        bad = ((a.reward != b.reward) | (a.done != b.done)).nonzero().flatten()
        assert len(bad) == 0, (t, len(bad), bad[:8].tolist(), a.reward[bad[:4]].tolist(), b.reward[bad[:4]].tolist(), a.done[bad[:4]].tolist(), b.done[bad[:4]].tolist())
        for k in a.obs:
            assert torch.equal(a.obs[k], b.obs[k]), (t, k)
    a.close(); b.close()


def test_presampled_action_is_never_stale():
    """The fused sampler draws the action of step t+1 at the end of the launch of step t and keeps it in the state
    header; whatever happens in between -- jumps in the step index, another seed, an explicit step, a masked reset,
    restoring an older state -- the action a fused launch takes must equal what pcbenv_sample_actions draws from the
    mask the launch starts from."""
    for name in ("c2", "c3", "c4"):
        cfg = named_config(name)
        B = 64
        env = BatchedPlacementEnv(cfg, B, queue_depth=2, run_seed=9, auto_reset=True)
        env.generate_instances(); env.reset()
        rng = np.random.RandomState(1)

        def fused(t):
            want = env.sample_actions(t).clone()
            _, _, _, _, got = env.rollout_step(t)
            assert torch.equal(got, want), (name, t)

        t = 0
        for _ in range(3):
            fused(t); t += 1                       # consecutive: the presampled action is used
        fused(t + 5); t += 6                       # step index jumps: tag mismatch -> fresh draw
        fused(t); t += 1
        env.run_seed += 1                          # another seed
        fused(t); t += 1
        fused(t); t += 1
        a = env.sample_actions(1234)               # explicit step in between clears the presampled action
        env.step(a)
        fused(t); t += 1
        fused(t); t += 1
        snap = env.state_dict()
        fused(t); fused(t + 1)
        env.load_state_dict(snap)                  # older state, with the action presampled for step t
        fused(t); t += 1
        mask = torch.from_numpy((rng.rand(B) < 0.5).astype(np.uint8)).to(env.device)
        env.reset(mask)                            # masked reset changes the mask under a presampled action
        fused(t); t += 1
        for _ in range(cfg.max_num_components + 2):  # across the in-launch resets
            fused(t); t += 1
        env.close()


@pytest.mark.parametrize("label,cfg_fn,B", [
    ("c3_centroid", lambda: named_config("c3"), 2048),
    ("c3_beam_k2", lambda: named_config("c3", "beam"), 1024),
    ("c3_both_k3", lambda: EnvConfig.pin(64, 64, 9, 9, 2, 6, 2, 6, 16, 16, 8, 8, 6, 6, "both", 3, 0.5), 1024),
    ("c4_centroid", lambda: named_config("c4"), 1024),
    ("pin_ragged_both_k4", lambda: EnvConfig.pin(48, 40, 7, 6, 2, 6, 2, 5, 14, 8, 3, 7, 9, 2, "both", 4, 0.3), 1024),
    # the beam search's register build at its widest (7-8 pins per net, k = 2) and at k = 1; its wide build with
    # nets beyond 8 pins; four wavefronts per environment (16 nets x 4 lanes = one wavefront's worth of search lanes)
    ("pin_8pins_beam_k2", lambda: EnvConfig.pin(64, 64, 9, 9, 2, 6, 2, 6, 16, 16, 4, 6, 8, 7, "beam", 2, 0.5), 512),
    ("pin_beam_k1", lambda: EnvConfig.pin(64, 64, 9, 9, 2, 6, 2, 6, 16, 16, 8, 8, 6, 3, "beam", 1, 0.5), 512),
    ("pin_12pins_both_k3", lambda: EnvConfig.pin(64, 64, 9, 9, 3, 6, 3, 6, 16, 16, 3, 4, 12, 9, "both", 3, 0.5), 512),
    ("c5_beam_k2", lambda: named_config("c5", "beam"), 128),
])
def test_reward_sweep_many_episodes(label, cfg_fn, B):
    """The routing reward is where rare geometry lives (shared end points, parallel segments, equal beam distances):
    thousands of episodes per configuration, reward / done / info of every step bit-exact against the oracle replaying
    the recorded actions.  PCBENV_SWEEP_SCALE repeats the sweep with further instance seeds."""
    cfg = cfg_fn()
    for rep in range(int(os.environ.get("PCBENV_SWEEP_SCALE", "1"))):
        _reward_sweep(label, cfg, B, episodes=4, run_seed=23 + rep)


def _reward_sweep(label, cfg, B, episodes, run_seed):
    from oracle import oracle as orc
    env = BatchedPlacementEnv(cfg, B, queue_depth=episodes + 1, run_seed=run_seed, auto_reset=True, incremental_obs=True)
    packed = env.generate_instances()
    env.reset()
    ob = orc.OracleBatch(cfg, B)
    ob.reset_packed(packed[0], np.ones(B, np.uint8))
    cursor = np.ones(B, np.int64)
    done_eps, t, n_routed = 0, 0, 0
    while done_eps < episodes * B and t < episodes * (cfg.max_num_components + 2):
        _, r, d, _, a = env.rollout_step(t)
        a = a.cpu().numpy(); r = r.cpu().numpy(); d = d.cpu().numpy(); inf = env.info_raw.cpu().numpy()
        rr, dd, ii = ob.step(a)
        assert np.array_equal(d, dd), (label, t)
        assert _same_bits(r, rr), (label, t, np.flatnonzero(r != rr)[:5], r[r != rr][:3], rr[r != rr][:3])
        has = ~np.isnan(inf[:, 0])
        assert _same_bits(inf[has], ii[has]), (label, t)
        n_routed += int(has.sum())
        fin = np.flatnonzero(dd)
        if len(fin):  # the GPU has already reset these from the next queue slot
            rec = np.stack([packed[cursor[i] % (episodes + 1)][i] for i in range(B)])
            ob.reset_packed(rec, dd.astype(np.uint8))
            cursor[fin] += 1
        done_eps += int(dd.sum())
        t += 1
    assert done_eps >= episodes * B and n_routed >= episodes * B // 2, (done_eps, n_routed)
    env.close()
