"""ctypes binding of libmirl.so — the C ABI declared in include/mi_rl.h.

The HIP library IS the product: there is no Python/CPU fallback.  Importing this module without a
built ``deep_rl_amd/libmirl.so`` raises, and every wrapper raises ``MiError`` on a non-zero return code.
Build with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C deep_rl_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("MIRL_SO", os.path.join(_HERE, "libmirl.so"))  # MIRL_SO: A/B builds of the same ABI

ABI_VERSION = 107   # == MI_VERSION of the include/mi_rl.h these signatures and struct layouts were written against
NPARAMS = 9155
DQN_NPARAMS = 10934
ACTOR_NPARAMS = 4610
DUELING_NPARAMS = 11019
SAC_Q_NPARAMS = 67329
SAC_ACTOR_NPARAMS = 67330
MI_OK = 0


class MiError(RuntimeError):
    pass


class Episode(C.Structure):
    _fields_ = [("env", C.c_int32), ("t", C.c_int32), ("ret", C.c_float), ("len", C.c_int32)]


class PPOBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "params", "exp_avg", "exp_avg_sq", "grads", "loss_terms", "grad_norm", "obs_cur", "observations", "values",
        "actions", "log_probs", "rewards", "dones", "advantages", "returns", "perm", "adv_sums", "workspace",
        "episodes", "episode_stats")] + [("max_ep", C.c_int32), ("episode_stats_next", C.c_void_p)]


class SacOwedAlpha(C.Structure):   # mi_sac_owed_alpha_t
    _fields_ = [("log_alpha", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("alpha", C.c_void_p), ("out", C.c_void_p),
                ("target_entropy", C.c_float), ("step", C.c_int64), ("lr", C.c_double), ("update_index", C.c_uint64), ("epoch", C.c_int32), ("stash_slot", C.c_int32)]


class SacCriticStep(C.Structure):   # mi_sac_critic_step_t
    _fields_ = [("workspace", C.c_void_p), ("batch", C.c_int32), ("q", C.c_void_p), ("q_target", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("grads", C.c_void_p), ("losses", C.c_void_p), ("step", C.c_int64), ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("adam_eps", C.c_double),
                ("tau", C.c_float)]


class PPOHparams(C.Structure):
    _fields_ = [("T", C.c_int32), ("n_minibatch", C.c_int32), ("update_epochs", C.c_int32), ("update_index", C.c_int32),
                ("opt_step", C.c_int64),
                ("gamma", C.c_float), ("gae_lambda", C.c_float), ("clip_coef", C.c_float), ("ent_coef", C.c_float),
                ("vf_coef", C.c_float), ("max_grad_norm", C.c_float),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double)]


# name -> (restype, argtypes); every symbol include/mi_rl.h declares
_VP, _I, _U64, _F, _D, _I64, _SZ, _U32 = C.c_void_p, C.c_int, C.c_uint64, C.c_float, C.c_double, C.c_int64, C.c_size_t, C.c_uint32
SIGNATURES = {
    "mi_version": (_I, []),
    "mi_last_error": (C.c_char_p, []),
    "mi_source_id": (C.c_char_p, []),
    "mi_env_create": (_I, [_I, _I, _U64, _U64, C.POINTER(_VP)]),
    "mi_env_destroy": (_I, [_VP]),
    "mi_env_reset": (_I, [_VP, _VP, _VP, _VP]),
    "mi_env_step": (_I, [_VP] * 10),
    "mi_env_step_ex": (_I, [_VP] * 11),
    "mi_env_get_state": (_I, [_VP, _VP, _VP, _VP]),
    "mi_ppo_forward": (_I, [_VP, _VP, _I, _VP, _VP, _VP]),
    "mi_ppo_rollout": (_I, [_VP, _VP, _I] + [_VP] * 12 + [_I, _VP]),
    "mi_gae": (_I, [_VP, _VP, _VP, _I, _I, _F, _F, _VP, _VP, _VP]),
    "mi_make_perm": (_I, [_U32, _U64, _VP, _VP]),
    "mi_perm_key": (_U64, [_U64, _U64, _U64]),
    "mi_adv_stats": (_I, [_VP, _VP, _I, _I, _VP, _VP]),
    "mi_ppo_workspace_bytes": (_SZ, []),
    "mi_ppo_set_contraction": (_I, [_I]),
    "mi_ppo_get_contraction": (_I, []),
    "mi_test_contraction": (_I, [_I, _VP, _VP, _I, _VP, _VP]),
    "mi_ppo_minibatch_grad": (_I, [_VP] * 8 + [_I, _VP, _F, _F, _F, _D, _VP, _VP, _VP, _VP]),
    "mi_clip_adam": (_I, [_VP, _VP, _VP, _VP, _I, _I64, _D, _D, _D, _D, _F, _VP, _VP]),
    "mi_explained_var": (_I, [_VP, _VP, _SZ, _VP, _VP]),
    "mi_explained_var_parts": (_I, [_VP, _VP, _SZ, _VP, _VP, _VP]),
    "mi_ppo_test_assume_sharded": (_I, [_I]),
    "mi_ppo_update": (_I, [_VP, C.POINTER(PPOBuffers), C.POINTER(PPOHparams), _VP]),
    "mi_ppo_update_sharded": (_I, [_VP, C.POINTER(PPOBuffers), C.POINTER(PPOHparams), _VP, _VP]),
    "mi_comm_unique_id": (_I, [_VP]),
    "mi_comm_create": (_I, [_VP, _I, _I, C.POINTER(_VP)]),
    "mi_comm_destroy": (_I, [_VP]),
    "mi_comm_info": (_I, [_VP, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "mi_comm_allreduce_sum": (_I, [_VP, _VP, _SZ, _I, _VP]),
    "mi_comm_p2p_alloc": (_I, [_I, _I, _SZ, C.POINTER(_VP), _VP]),
    "mi_comm_p2p_connect": (_I, [_VP, _VP]),
    "mi_comm_p2p_synthetic": (_I, [_I, _SZ, C.POINTER(_VP)]),
    "mi_comm_check": (_I, [_VP]),
    "mi_comm_carrier": (_I, [_VP]),
    "mi_comm_p2p_set_colocated": (_I, [_VP, _I]),
    "mi_comm_p2p_set_fused": (_I, [_VP, _I]),
    "mi_comm_poll": (_I, [_VP]),
    "mi_comm_test_set_seq": (_I, [_VP, _U32]),
    "mi_dqn_forward": (_I, [_VP, _VP, _I, _VP, _VP]),
    "mi_dqn_act_steps": (_I, [_VP, _VP, _I, _I64, _I64, _I64, _D, _D, _D, _I64] + [_VP] * 9 + [_I, _VP]),
    "mi_dqn_act_steps2": (_I, [_VP, _VP, _I, _I64, _I64, _I64, _D, _D, _D, _I64] + [_VP] * 9 + [_I, _VP, _VP]),
    "mi_dqn_sample": (_I, [_U64, _U64, _I64, _I, _VP, _VP]),
    "mi_dqn_workspace_bytes": (_SZ, [_I]),
    "mi_dqn_td_grad": (_I, [_VP] * 7 + [_I, _I, _I64, _F, _D, _VP, _VP, _VP, _VP]),
    "mi_ppo_perms_and_stats": (_I, [_U64, _I, _I, _I, _I, _VP, _VP, _VP, _VP]),
    "mi_ppo_rollout_gae": (_I, [_VP, _VP, _I] + [_VP] * 9 + [_I, _F, _F, _VP, _VP, _VP]),
    "mi_env_episode_stats": (_I, [_VP, _VP, _VP]),
    "mi_env_state_bytes": (_SZ, [_VP]),
    "mi_env_export_state": (_I, [_VP, _VP, _VP]),
    "mi_env_import_state": (_I, [_VP, _VP, _VP]),
    "mi_dqn_td_update": (_I, [_VP] * 7 + [_I, _I, _I64, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I64, _D, _D, _D, _D, _U64, _U64, _I64, _VP]),
    "mi_dqn_td_update_sharded": (_I, [_VP] * 7 + [_I, _I, _I64, _F, _VP, _VP, _VP, _VP, _VP, _VP, _I64, _D, _D, _D, _D, _F, _VP, _U64, _U64, _I64, _VP, _VP]),
    "mi_sac_critic_update_sharded": (_I, [_VP] * 8 + [_I, _I, _I64, _VP, _U64, _U64, _VP, _F, _VP, _VP, _VP, _VP, _I64, _D, _D, _D, _D, _F, _VP, _VP]),
    "mi_sac_actor_update_sharded": (_I, [_VP] * 4 + [_I, _VP, _U64, _U64, _VP, _VP, _VP, _VP, _VP, _I64, _D, _D, _D, _D, _VP, _VP]),
    "mi_sac_alpha_step_sharded": (_I, [_VP, _VP, _VP, _I, _VP, _U64, _U64, _F, _VP, _VP, _VP, _I64, _D, _VP, _VP, _VP, _VP, _VP, _VP]),
    "mi_per_workspace_bytes": (_SZ, [_I64]),
    "mi_per_mark": (_I, [_VP, _I, _I64, _I64, _I, _VP, _VP]),
    "mi_per_sample": (_I, [_U64, _U64, _VP, _I64, _I64, _D, _F, _F, _I, _I, _VP, _VP, _VP, _VP]),
    "mi_per_td_grad": (_I, [_VP] * 7 + [_I, _I, _I64, _F, _D, _VP, _VP, _VP, _VP, _VP, _VP]),
    "mi_per_update_priorities": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _VP]),
    "mi_per_sums_refresh": (_I, [_VP, _I64, _F, _VP, _VP]),
    "mi_per_mark_sums": (_I, [_VP, _I, _I64, _I64, _I, _VP, _F, _VP, _VP]),
    "mi_per_sample_current": (_I, [_U64, _U64, _VP, _I64, _I64, _D, _F, _F, _I, _I, _VP, _VP, _VP, _VP]),
    "mi_per_update_priorities_sums": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _I64, _F, _VP, _VP]),
    "mi_per_act_steps": (_I, [_VP, _VP, _I, _I64, _I64, _I64, _D, _D, _D, _I64] + [_VP] * 9 + [_I, _VP, _VP, _VP, _F, _VP, _VP, _I, _VP]),
    "mi_per_td_update": (_I, [_VP] * 7 + [_I, _I, _I64, _F] + [_VP] * 7 + [_I64, _D, _D, _D, _D, _U64, _U64, _VP, _I64, _D, _F, _F, _I, _VP, _VP, _VP, _VP]),
    "mi_per_settle_sums": (_I, [_VP, _VP, _I, _I64, _F, _VP, _VP]),
    "mi_dueling_pack": (_I, [_VP, _VP, _VP]),
    "mi_dueling_unpack_grads": (_I, [_VP, _VP, _VP]),
    "mi_dueling_td_update": (_I, [_VP] * 7 + [_I, _I, _I64, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I64, _D, _D, _D, _D, _U64, _U64, _I64, _VP]),
    "mi_env_step_cont": (_I, [_VP] * 10),
    "mi_sac_actor_sample": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _VP]),
    "mi_sac_q_forward": (_I, [_VP, _VP, _VP, _I, _VP, _VP]),
    "mi_sac_act_step": (_I, [_VP, _VP, _I64, _I64, _I64] + [_VP] * 10 + [_I, _VP]),
    "mi_sac_workspace_bytes": (_SZ, [_I]),
    "mi_sac_critic_grad": (_I, [_VP] * 8 + [_I, _I, _I64, _VP, _U64, _U64, _VP, _F, _D, _VP, _VP, _VP, _VP]),
    "mi_sac_actor_grad": (_I, [_VP] * 4 + [_I, _VP, _U64, _U64, _VP, _D, _VP, _VP, _VP, _VP]),
    "mi_sac_critic_update": (_I, [_VP] * 8 + [_I, _I, _I64, _VP, _U64, _U64, _VP, _F, _VP, _VP, _VP, _VP, _VP, _I64, _D, _D, _D, _D, _F, _U64, _I64, _VP]),
    "mi_sac_actor_update": (_I, [_VP] * 4 + [_I, _VP, _U64, _U64, _VP, _VP, _VP, _VP, _VP, _VP, _I64, _D, _D, _D, _D, _VP]),
    "mi_sac_critic_update_owed": (_I, [_VP] * 8 + [_I, _I, _I64, _VP, _U64, _U64, _VP, _F, _VP, _VP, _VP, _VP, _VP, _I64, _D, _D, _D, _D, _F, _U64, _I64, _VP, _VP]),
    "mi_sac_actor_update_owed": (_I, [_VP] * 4 + [_I, _VP, _U64, _U64, _VP, _VP, _VP, _VP, _VP, _VP, _I64, _D, _D, _D, _D, _VP, _VP]),
    "mi_sac_alpha_step_owed": (_I, [_VP, _I, _U64, _VP, _VP, _VP]),
    "mi_sac_critic_update_deferred": (_I, [_VP] * 8 + [_I, _I, _I64, _VP, _U64, _U64, _VP, _F, _VP, _U64, _I64, _VP, _VP]),
    "mi_sac_critic_step": (_I, [_VP, _VP]),
    "mi_sac_act_step_carry": (_I, [_VP, _VP, _I64, _I64, _I64] + [_VP] * 10 + [_I, _VP, _VP]),
    "mi_sac_owed_alpha_fits": (_I, [_I]),
    "mi_sac_shadow_set": (_I, [_VP, _I, _VP]),
    "mi_sac_shadow_refresh": (_I, [_VP, _VP]),
    "mi_sac_shadow_invalidate": (_I, [_VP]),
    "mi_sac_shadow_valid": (_I, [_VP]),
    "mi_sac_check": (_I, [_VP, _I]),
    "mi_sac_clear_error": (_I, [_VP, _I, _VP]),
    "mi_sac_set_max_cus": (_I, [_I]),
    "mi_sac_usable_cus": (_I, []),
    "mi_sac_test_fault": (_I, [_I]),
    "mi_sac_alpha_step": (_I, [_VP, _VP, _VP, _I, _VP, _U64, _U64, _F, _VP, _VP, _VP, _I64, _D, _VP, _VP, _VP, _VP]),
    "mi_sac_mean_logp": (_I, [_VP, _VP, _VP, _I, _VP, _U64, _U64, _D, _VP, _VP, _VP]),
    "mi_sac_alpha_adam": (_I, [_VP, _F, _VP, _VP, _VP, _I64, _D, _VP, _VP, _VP]),
    "mi_adam": (_I, [_VP] * 4 + [_I, _I64, _D, _D, _D, _D, _VP]),
    "mi_polyak": (_I, [_VP, _VP, _I, _F, _VP]),
    "mi_selftest_mfma": (_I, [_VP, _VP, _VP]),
    "mi_test_tanh": (_I, [_VP, _VP, _I, _VP]),
    "mi_prof_begin": (_I, [_I, _U32]),
    "mi_prof_end": (_I, [C.POINTER(_F), C.POINTER(C.c_int32)]),
    "mi_prof_pause": (_I, [_I]),
    "mi_timer_create": (_I, [C.POINTER(_VP)]),
    "mi_timer_destroy": (_I, [_VP]),
    "mi_timer_start": (_I, [_VP, _VP]),
    "mi_timer_stop": (_I, [_VP, _VP]),
    "mi_timer_elapsed_ms": (_I, [_VP, C.POINTER(_F)]),
}

_lib = None


def lib():
    """Load libmirl.so (once).  Fails loudly: the HIP extension is not optional."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise MiError("deep_rl_amd: %s is missing — build it with `make -C deep_rl_amd/csrc` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % SO_PATH)
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        got = L.mi_version()
        if got != ABI_VERSION:   # a stale libmirl.so would read structs of another layout: refuse instead of corrupting memory
            raise MiError("deep_rl_amd: %s reports ABI version %d, this binding is written against %d — rebuild it (make -C deep_rl_amd/csrc)" % (SO_PATH, got, ABI_VERSION))
        _lib = L
        mode = os.environ.get("MIRL_PPO_CONTRACTION", "f32")   # experiment switch, see set_contraction
        if mode != "f32":
            set_contraction(mode)
    return _lib


CONTRACTIONS = ("f32", "bf16x3")   # == enum MI_CONTRACTION_* of include/mi_rl.h


def set_contraction(mode):
    """Which matrix pipe the PPO gradient kernel's 64 x 64 contractions run on, process-wide: "f32" (default, exact f32 MFMA) or "bf16x3"
    (experiment: three-part bf16 split of both operands, six products, f32 accumulate — f32-grade, not bit-identical).  Also settable with
    the environment variable MIRL_PPO_CONTRACTION."""
    if mode not in CONTRACTIONS:
        raise MiError("unknown contraction %r (known: %s)" % (mode, ", ".join(CONTRACTIONS)))
    check(lib().mi_ppo_set_contraction(CONTRACTIONS.index(mode)), "mi_ppo_set_contraction")


def get_contraction():
    return CONTRACTIONS[lib().mi_ppo_get_contraction()]


def check(rc, what=""):
    if rc != MI_OK:
        msg = lib().mi_last_error()
        raise MiError("%s failed (rc=%d): %s" % (what or "libmirl call", rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "libmirl needs contiguous tensors"
    return t.data_ptr()


def stream_ptr(device=None):
    """The raw hipStream_t of torch's current stream, so kernels order with torch / RCCL work."""
    import torch

    return torch.cuda.current_stream(device).cuda_stream


PROF_TAGS = ("rollout", "gae", "grad", "reduce", "clip_adam", "stats", "dqn_act", "dqn_td", "dqn_reduce", "per",
             "sac_act", "sac_critic", "sac_actor", "sac_gemm", "sac_assemble", "sac_logp", "comm_grad", "comm_stats")   # == enum MI_PROF_* of include/mi_rl.h


def prof_begin(max_launches, tags=None):
    """Arm the in-library profiler for `tags` (names from PROF_TAGS; default all)."""
    mask = 0
    for t in (tags or PROF_TAGS):
        mask |= 1 << PROF_TAGS.index(t)
    check(lib().mi_prof_begin(int(max_launches), mask), "mi_prof_begin")


def prof_pause(paused):
    """Stop / resume the sampling between prof_begin and prof_end."""
    check(lib().mi_prof_pause(1 if paused else 0), "mi_prof_pause")


def prof_end():
    """-> {tag: (total_ms, launches)} of every tagged kernel launched since prof_begin (synchronises)."""
    ms = (_F * len(PROF_TAGS))()
    cnt = (C.c_int32 * len(PROF_TAGS))()
    check(lib().mi_prof_end(ms, cnt), "mi_prof_end")
    return {t: (ms[i], cnt[i]) for i, t in enumerate(PROF_TAGS)}


class Timer:
    """HIP-event timer on an explicit stream (torch.cuda.Event only sees torch's current stream)."""

    def __init__(self):
        self.h = _VP()
        check(lib().mi_timer_create(C.byref(self.h)), "mi_timer_create")

    def start(self, stream):
        check(lib().mi_timer_start(self.h, stream), "mi_timer_start")

    def stop(self, stream):
        check(lib().mi_timer_stop(self.h, stream), "mi_timer_stop")

    def elapsed_ms(self):
        ms = _F()
        check(lib().mi_timer_elapsed_ms(self.h, C.byref(ms)), "mi_timer_elapsed_ms")
        return ms.value

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mi_timer_destroy(self.h)
            self.h = None
