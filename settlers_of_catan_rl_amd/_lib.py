"""ctypes binding of libcatan_hip.so (the C ABI declared in include/catan_hip.h; the net's kernels in catan_hip_nn.h, knobs and
profilers in catan_hip_tuning.h).

There is NO CPU fallback: if the shared library is missing or there is no HIP device the calls raise.
"""
import ctypes as C
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libcatan_hip.so")


class CatanHipError(RuntimeError):
    pass


class CatanWgradProblem(C.Structure):
    """catan_wgrad_problem_t (include/catan_hip_nn.h)"""
    _fields_ = [("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("rows", C.c_int64),
                ("in_features", C.c_int32), ("out_features", C.c_int32), ("dw_ld", C.c_int32), ("dw_col0", C.c_int32)]


class CatanCfg(C.Structure):
    _fields_ = [("max_proposed_trades_per_turn", C.c_int32), ("dense_reward", C.c_int32), ("validate_actions", C.c_int32),
                ("auto_reset", C.c_int32), ("win_reward", C.c_double), ("reward_annealing_factor", C.c_double),
                ("max_actions_per_turn", C.c_int32), ("reserved_", C.c_int32)]


def _sources():
    out = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h", ".inc"))]
    inc = os.path.join(os.path.dirname(PKG_DIR), "include")
    out += [os.path.join(inc, f) for f in sorted(os.listdir(inc)) if f.endswith(".h")]
    return out


_HASH_MARK = b"CATAN_BUILD_HASH="


# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (gfx950 has one register file; the default allocates them to the AGPR half and
# copies every element back with v_accvgpr_read before the VALU may touch it: 432 of k_attn_mfma_bwd's ~1 900 instructions)
# -fno-slp-vectorize: no packed-f32 VALU (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) anywhere.  They issue at the rate of the two
# scalar instructions they replace (MI355X_MICROARCH.md), and next to v_cvt_pk_bf16_f32 they produced a timing-dependent wrong result:
# k_tile_encoder_fwd's LayerNorm with its bf16 pairs converted by one v_cvt_pk_bf16_f32 each gave, with two workgroups per CU, wrong
# rows for the tokens of a wave's lanes 48..63 in ~0.6 % of the boards, differently on every run (the same source built
# with this flag is bit-stable, and tests/test_gpu_ppo_pipeline.py::test_fused_tile_encoder_forward_vs_unfused is the check)
# -target-feature -packed-fp32-ops (round 5): the SLP flag only stops ONE source of those instructions; the load / store vectoriser and the
# DAG combiner produced them too (llvm-objdump of the round-4 library: 12 v_pk_mul_f32 in each k_obs_rows<fp32>, one v_pk_add_f32 beside three
# v_cvt_pk_bf16_f32 in k_lnw_bwd<bf16, 2, 64> - the very combination that misbehaved).  With the target feature off the back end cannot
# select them at all, whoever asks; `check_no_packed_f32` disassembles every freshly built library and refuses one that has any.
# The flag is NOT a no-op, although the build prints "'-packed-fp32-ops' is not a recognized feature for this target (ignoring feature)":
# -Xclang reaches both halves of the compilation, the x86 HOST half does not know an AMDGPU feature and says so (LLVM's subtarget parser,
# not a clang diagnostic: no -Wno- switch, and -Xarch_device refuses to forward -Xclang), while the gfx950 half honours it -
# `packed_fp32_flag_effect()` below compiles a two-line kernel both ways (v_pk_fma_f32 without the flag, two v_fma_f32 with it;
# tests/test_host_logic_cpu.py::test_packed_fp32_feature_flag_is_not_a_no_op).  The compiler's output is printed as it comes.
BUILD_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-fno-slp-vectorize",
               "-mllvm", "-amdgpu-mfma-vgpr-form", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
LLVM_OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def device_code_object(path=None):
    """The gfx950 code object embedded in the library (clang offload bundle: magic, entry table, blobs) -> bytes"""
    import struct
    with open(LIB_PATH if path is None else path, "rb") as f:
        blob = f.read()
    i = blob.find(b"__CLANG_OFFLOAD_BUNDLE__")
    if i < 0:
        raise CatanHipError("no offload bundle in the library")
    n = struct.unpack_from("<Q", blob, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", blob, off)
        triple = blob[off + 24:off + 24 + tl].decode()
        off += 24 + tl
        if "gfx950" in triple:
            return blob[i + o:i + o + sz]
    raise CatanHipError("no gfx950 code object in the library")


def check_no_packed_f32(path=None):
    """Disassembles the library's device code and raises if any packed-fp32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)
    is in it: next to v_cvt_pk_bf16_f32 they gave a timing-dependent wrong result in k_tile_encoder_fwd, and the packed
    conversion is used by every bf16 kernel of the net.  -> the number of v_cvt_pk_bf16_f32 found (informational); None without llvm-objdump."""
    import re
    import tempfile
    if not os.path.exists(LLVM_OBJDUMP):
        return None
    with tempfile.NamedTemporaryFile(suffix=".co", dir="/tmp") as tmp:
        tmp.write(device_code_object(path))
        tmp.flush()
        text = subprocess.run([LLVM_OBJDUMP, "-d", tmp.name], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode("ascii", "replace")
    fn, bad = "?", {}
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            fn = m.group(1)
        elif re.search(r"\bv_pk_(fma|mul|add)_f32\b", line):
            bad[fn] = bad.get(fn, 0) + 1
    if bad:
        raise CatanHipError("packed-fp32 VALU instructions in the library (see _lib.BUILD_FLAGS): " + ", ".join(f"{k}: {v}" for k, v in sorted(bad.items())[:8]))
    return len(re.findall(r"\bv_cvt_pk_bf16_f32\b", text))


def packed_fp32_flag_effect(workdir="/tmp"):
    """Compiles a two-line kernel (a float2 multiply-add) for gfx950 without and with the `-target-feature -packed-fp32-ops` pair of BUILD_FLAGS
    -> (packed instructions without the flag, with the flag).  Evidence that the flag acts on the device half (cross-compiles: no GPU needed)."""
    import re
    import tempfile
    src = ("#include <hip/hip_runtime.h>\n__global__ void k(const float2* a, const float2* b, float2* c, int n) {\n"
           "  int i = blockIdx.x * blockDim.x + threadIdx.x;\n"
           "  if (i < n) { float2 x = a[i], y = b[i]; c[i] = make_float2(x.x * y.x + 1.0f, x.y * y.y + 1.0f); }\n}\n")
    counts = []
    with tempfile.TemporaryDirectory(dir=workdir) as d:
        with open(os.path.join(d, "t.hip"), "w") as f:
            f.write(src)
        for extra in ([], ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]):
            subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "--cuda-device-only", "-S"] + extra + ["t.hip", "-o", "t.s"], cwd=d, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            with open(os.path.join(d, "t.s")) as f:
                counts.append(len(re.findall(r"\bv_pk_(?:fma|mul|add)_f32\b", f.read())))
    return tuple(counts)


def source_hash():
    """sha256 over the names and contents of csrc/* and include/*.h and the compiler flags: what the library is built from."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(BUILD_FLAGS).encode() + b"\0")
    for path in _sources():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:32]


def binary_hash(path=None):
    """The source hash baked into a built library (`catan_build_hash()`), read from the file without loading it."""
    path = LIB_PATH if path is None else path
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        blob = f.read()
    i = blob.find(_HASH_MARK)
    if i < 0:
        return None
    return blob[i + len(_HASH_MARK):i + len(_HASH_MARK) + 32].decode("ascii", "replace")


def build_library(force=False, verbose=False):
    """hipcc cross-compiles for gfx950 without a GPU; the .so stays in-tree (it travels with the repo snapshot).  Rebuilt
    whenever the hash of the sources differs from the one baked into the binary - file times mean nothing on a fresh checkout
    or on a snapshot copied to the GPU box."""
    want = source_hash()
    cmd = ["hipcc"] + BUILD_FLAGS + [f'-DCATAN_BUILD_HASH_STR="{want}"', "-o", LIB_PATH, os.path.join(CSRC, "catan_abi.hip")]
    have = binary_hash()
    if not force and have == want:
        if verbose:
            print(f"libcatan_hip.so carries the hash of its {len(_sources())} sources ({want}): not recompiled (command: {' '.join(cmd)})")
        return LIB_PATH
    if verbose:
        print(f"compiling (binary hash {have}, source hash {want}): " + " ".join(cmd))
    r = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = r.stdout.decode("utf-8", "replace")
    if out.strip():
        print(out.rstrip())
        if "'-packed-fp32-ops' is not a recognized feature" in out:
            print("(those notes come from the x86 host half of the compilation; the gfx950 half honours the feature: _lib.packed_fp32_flag_effect, check_no_packed_f32)")
    if r.returncode != 0:
        raise CatanHipError(f"hipcc failed with exit code {r.returncode}")
    if binary_hash() != want:
        raise CatanHipError("the freshly built libcatan_hip.so does not carry the source hash")
    n_cvt = check_no_packed_f32()
    if verbose and n_cvt is not None:
        print(f"disassembly: no v_pk_*_f32, {n_cvt} v_cvt_pk_bf16_f32")
    if verbose:
        print(f"built {LIB_PATH} ({os.path.getsize(LIB_PATH)} bytes)")
    return LIB_PATH


_lib = None
_vp = C.c_void_p

_SIGS = {
    "catan_build_hash": (C.c_char_p, []),
    "catan_cfg_default": (None, [C.POINTER(CatanCfg)]),
    "catan_state_words": (C.c_int32, []),
    "catan_mask_words": (C.c_int32, []),
    "catan_action_words": (C.c_int32, []),
    "catan_obs_floats": (C.c_int32, []),
    "catan_state_bytes_per_game": (C.c_int32, []),
    "catan_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int64, C.c_uint64, C.c_uint64, C.POINTER(CatanCfg)]),
    "catan_destroy": (None, [_vp]),
    "catan_last_error": (C.c_char_p, []),
    "catan_num_envs": (C.c_int64, [_vp]),
    "catan_reset": (C.c_int, [_vp, _vp, _vp]),
    "catan_reset_board_only": (C.c_int, [_vp, _vp]),
    "catan_seed_mt19937": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp]),
    "catan_mt19937_set_state": (C.c_int, [_vp, C.c_int32, _vp, C.c_int32, _vp]),
    "catan_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "catan_step_deferred": (C.c_int, [_vp, _vp, C.c_int32, _vp, _vp, _vp, _vp]),
    "catan_step_flush": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "catan_masks": (C.c_int, [_vp, _vp, _vp]),
    "catan_masks_packed": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_int64)]),
    "catan_expand_masks": (C.c_int, [_vp, C.c_int64, C.c_int32, _vp, _vp]),
    "catan_masks_packed_copy": (C.c_int, [_vp, _vp, _vp]),
    "catan_masked_row_store": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, _vp]),
    "catan_ffn_bwd_dx": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "catan_ffn_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "catan_ffn_outproj_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "catan_ffn_outproj_bwd_rh": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "catan_qkv_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "catan_qkv_bwd_dx": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, C.c_int64, _vp]),
    "catan_weight_image_bytes": (C.c_int32, []),
    "catan_weight_images": (C.c_int, [_vp, C.c_int32, _vp]),
    "catan_recurrent_given": (C.c_int, [_vp, C.c_int64, _vp, _vp, C.c_int32, C.c_int32, C.c_int64, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "catan_categorical_bits_fwd": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp, C.c_int64, C.c_int, _vp]),
    "catan_categorical_bits_bwd": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int, _vp]),
    "catan_wgrad_big_workspace_floats": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "catan_linear_wgrad_big": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_int, _vp]),
    "catan_adam_chunk_elements": (C.c_int32, []),
    "catan_adam_step": (C.c_int, [_vp, _vp, C.c_int32, _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _vp, _vp]),
    "catan_gather_rows": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int64, C.c_int64, _vp]),
    "catan_expand_rows": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_int64, _vp]),
    "catan_segment_sum_rows": (C.c_int, [_vp, C.c_int64, _vp, _vp, C.c_int64, _vp, C.c_int64, _vp]),
    "catan_concat_rows": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int64, C.c_int64, _vp]),
    "catan_scatter_rows_ranges": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int, _vp, _vp, _vp, C.c_int64, _vp]),
    "catan_collector_pre": (C.c_int, [C.c_int64, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "catan_collector_post": (C.c_int, [C.c_int64, C.c_int32] + [_vp] * 23),
    "catan_deciding_seat": (C.c_int, [_vp, _vp, _vp]),
    "catan_sample_random_actions": (C.c_int, [_vp, C.c_uint32, _vp, _vp]),
    "catan_state_export": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp]),
    "catan_state_import": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp]),
    "catan_set_reward_annealing": (C.c_int, [_vp, C.c_double]),
    "catan_set_reward_f64_buffer": (C.c_int, [_vp, _vp]),
    "catan_invalid_action_count": (C.c_int64, [_vp, _vp]),
    "catan_random_rollout": (C.c_int, [_vp, C.c_uint32, C.c_int64, _vp]),
    "catan_obs": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "catan_obs_rows": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "catan_obs_rows_of": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "catan_masks_of": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp]),
    "catan_longest_path": (C.c_int, [_vp, _vp, _vp, _vp]),
    "catan_gae_workspace_doubles": (C.c_int64, [C.c_int64]),
    "catan_gae": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int64, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    "catan_adv_normalise": (C.c_int, [_vp, C.c_int64, _vp, _vp]),
    "catan_ppo_loss_workspace_doubles": (C.c_int64, []),
    "catan_ppo_loss": (C.c_int, [_vp] * 6 + [C.c_int64, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp]),
    "catan_attention_fwd": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "catan_attention_bwd": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "catan_layer_norm_fwd": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_float, C.c_int, C.c_int, _vp]),
    "catan_layer_norm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_float, C.c_int, C.c_int, _vp]),
    "catan_layer_norm_bwd_res": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_float, C.c_int, C.c_int, _vp]),
    "catan_profile_enable": (C.c_int, [_vp, C.c_int]),
    "catan_profile_read": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "catan_random_rollout_timed": (C.c_int, [_vp, C.c_uint32, C.c_int64, C.c_int32, _vp, C.POINTER(C.c_float)]),
    "catan_random_rollout_deferred": (C.c_int, [_vp, C.c_int64, C.c_int32, _vp]),
    "catan_policy_counters": (C.c_int, [_vp, _vp, _vp]),
    "catan_set_policy_counters": (C.c_int, [_vp, _vp, _vp]),
    "catan_set_lr_budgets": (C.c_int, [_vp, C.c_int32, C.c_int32]),
    "catan_set_deferred_fused": (C.c_int, [_vp, C.c_int32]),
    "catan_step_algorithmic_bytes": (C.c_int32, []),
    "catan_step_fused_algorithmic_bytes": (C.c_int32, []),
    "catan_hip_runtime_version": (C.c_int32, []),
    "catan_deferred_fused": (C.c_int32, [_vp]),
    "catan_set_step_wave_games": (C.c_int, [_vp, C.c_int32]),
    "catan_set_lr_rounds": (C.c_int, [_vp, C.c_int32, C.c_int32]),
    "catan_slow_path_counts": (C.c_int, [_vp, _vp, C.POINTER(C.c_uint64)]),
    "catan_profile_read_waves": (C.c_int, [_vp, _vp]),
    "catan_linear_rows_supported": (C.c_int, [C.c_int64, C.c_int, C.c_int]),
    "catan_linear_rows": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_int, _vp]),
    "catan_linear_rows_fused": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_int, _vp, C.c_int, _vp]),
    "catan_categorical_fwd": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int, _vp]),
    "catan_categorical_bwd": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int, _vp]),
    "catan_lstm_cell_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_int, _vp]),
    "catan_lstm_cell_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_int, _vp]),
    "catan_calib_copy": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "catan_tile_encoder_weight_elems": (C.c_int32, []),
    "catan_tile_encoder_vec_elems": (C.c_int32, []),
    "catan_tile_encoder_fwd": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "catan_tile_encoder_fwd_train": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, _vp, C.c_int64, _vp]),
    "catan_head_weight_elems": (C.c_int32, []),
    "catan_head_vec_elems": (C.c_int32, []),
    "catan_head_fwd": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64, C.c_int32, _vp, _vp, C.c_float, C.c_int32, _vp, C.c_int64, _vp, _vp, _vp, C.c_int64, _vp]),
    "catan_head_state_floats": (C.c_int32, []),
    "catan_head_chain": (C.c_int, [_vp, C.c_int64, _vp, _vp, C.c_float, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "catan_card_summary_params": (C.c_int32, []),
    "catan_card_summary_patterns": (C.c_int32, []),
    "catan_card_pattern_sum": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, C.c_int64, _vp]),
    "catan_card_summary_fwd": (C.c_int, [_vp, C.c_int, C.c_int64, _vp, _vp, C.c_float, _vp, _vp, C.c_int64, _vp]),
    "catan_card_summary_lookup": (C.c_int, [_vp, C.c_int, C.c_int64, _vp, _vp, _vp, C.c_float, _vp, C.c_int64, _vp]),
    "catan_card_summary_bwd": (C.c_int, [_vp, C.c_int, C.c_int64, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "catan_randomise_uncertainty": (C.c_int, [_vp, _vp, _vp]),
    "catan_players_turn_sim": (C.c_int, [_vp, _vp, _vp]),
    "catan_inconsistent_deal_count": (C.c_int64, [_vp, _vp]),
    "catan_missed_speculation_count": (C.c_int64, [_vp, _vp]),
    "catan_linear_wgrad_supported": (C.c_int, [C.c_int64, C.c_int, C.c_int]),
    "catan_linear_wgrad": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_int, _vp]),
    "catan_linear_wgrad_grouped": (C.c_int, [C.POINTER(CatanWgradProblem), C.c_int32, _vp]),
}


def declared_symbols():
    """Every entry point include/*.h declares (the CPU test-suite checks the built .so exports them)."""
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CatanHipError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(the HIP path has no CPU fallback)")
        # torch first: libcatan_hip.so must bind to the SAME libamdhip64 instance torch loaded, otherwise the
        # process ends up with two HIP runtimes and device pointers / streams cannot be shared.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        have, want = L.catan_build_hash().decode(), source_hash()
        if have != want and not os.environ.get("CATAN_ALLOW_STALE_LIB"):
            raise CatanHipError(f"{LIB_PATH} was built from other sources (binary {have}, sources {want}): rebuild with "
                                "`python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise CatanHipError(f"libcatan_hip: error {rc}: {lib().catan_last_error().decode()}")
