"""ctypes binding of libazamd.so (the C ABI declared in include/azamd.h).

The library is the product: there is no Python or CPU fallback for anything it
computes.  Loading fails loudly when the shared object is missing.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AZAMD_LIB") or os.path.join(_HERE, "libazamd.so")   # AZAMD_LIB: a diagnostic build of the same library


class AzError(RuntimeError):
    pass


class EngineConfig(C.Structure):
    _fields_ = [
        ("n_games", C.c_int32),
        ("c_puct", C.c_double),
        ("dirichlet_alpha", C.c_double),
        ("dirichlet_epsilon", C.c_double),
        ("temperature_threshold", C.c_int32),
        ("pool_nodes_per_game", C.c_int64),
        ("seed", C.c_uint64),
        ("device", C.c_int32),
    ]


class NetConfig(C.Structure):
    _fields_ = [
        ("num_blocks", C.c_int32),
        ("num_filters", C.c_int32),
        ("se_ratio", C.c_int32),
        ("max_batch", C.c_int32),
        ("device", C.c_int32),
    ]


class SelfplayStats(C.Structure):
    _fields_ = [
        ("simulations", C.c_uint64),
        ("nn_evals", C.c_uint64),
        ("moves", C.c_uint64),
        ("games_finished", C.c_uint64),
        ("samples_pending", C.c_uint64),
        ("network_rows", C.c_uint64),
    ]


# name -> (argtypes); every entry point returns int unless listed in _RESTYPES
_VP = C.c_void_p
_SIGNATURES = {
    "az_version": [],
    "az_movegen_encode": [C.c_int64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP],
    "az_engine_create": [C.POINTER(_VP), C.POINTER(EngineConfig)],
    "az_engine_destroy": [_VP],
    "az_engine_set_positions": [_VP, C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP],
    "az_engine_update_positions": [_VP, C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP],
    "az_engine_root_counts": [_VP, _VP, _VP],
    "az_engine_get_positions": [_VP, _VP, _VP, _VP, _VP, _VP],
    "az_engine_set_active": [_VP, _VP, _VP],
    "az_engine_roots_prepare": [_VP, _VP, _VP, _VP],
    "az_engine_roots_expand": [_VP, _VP, _VP],
    "az_engine_add_noise": [_VP, _VP, _VP],
    "az_engine_select": [_VP, _VP, _VP, _VP],
    "az_engine_expand_backup": [_VP, _VP, _VP, _VP],
    "az_engine_node_children": [_VP, C.c_int32, C.c_int64] + [_VP] * 11,
    "az_engine_action_probs": [_VP, _VP, _VP, _VP],
    "az_engine_roots_children": [_VP, C.c_int32] + [_VP] * 12,
    "az_engine_advance": [_VP, _VP, _VP],
    "az_engine_set_root": [_VP, C.c_int32, C.c_int64, _VP],
    "az_engine_set_roots": [_VP, C.c_int32, _VP, _VP, _VP],
    "az_engine_search_leaf_parallel": [_VP, C.c_int32, _VP, C.c_int32, C.c_int32, _VP],
    "az_engine_import_tree": [_VP, C.c_int32, C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP],
    "az_engine_pool_usage": [_VP, _VP, _VP],
    "az_engine_stats": [_VP, _VP, _VP, _VP],
    "az_net_create": [C.POINTER(_VP), C.POINTER(NetConfig)],
    "az_net_destroy": [_VP],
    "az_net_max_batch": [_VP],
    "az_net_load_state": [_VP, _VP, C.c_int64],
    "az_net_forward": [_VP, C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP, _VP],
    "az_net_forward_noalloc": [_VP, C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP, _VP],
    "az_net_forward_planes": [_VP, C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP],
    "az_net_set_timing": [_VP, C.c_int32],
    "az_net_set_tower_form": [_VP, C.c_int32],
    "az_net_debug_stamps": [_VP, _VP, C.c_int32],
    "az_net_tower_time": [_VP, C.POINTER(C.c_double), C.POINTER(C.c_int64)],
    "az_engine_attach_net": [_VP, _VP],
    "az_engine_search": [_VP, C.c_int32, C.c_int32, _VP],
    "az_engine_play_move": [_VP, C.c_int32, C.c_int32, _VP, C.c_int32, _VP],
    "az_engine_chosen_actions": [_VP, _VP, _VP],
    "az_engine_drain_samples": [_VP, C.c_int64, _VP, _VP, _VP, C.POINTER(C.c_int64), _VP],
    "az_engine_drain_packed": [_VP, C.c_int64, _VP, C.POINTER(C.c_int64), _VP],
    "az_unpack_samples": [C.c_int64, _VP, _VP, _VP, _VP, _VP],
    "az_engine_selfplay_stats": [_VP, C.POINTER(SelfplayStats), _VP],
}
_OPTIONAL = {}

PACKED_SAMPLE_BYTES = 792     # AZ_PACKED_SAMPLE_BYTES

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AzError(
                f"{LIB_PATH} is missing: build it with alphazero_amd/csrc/build.sh "
                "(__graft_entry__.build()).  There is no fallback path.")
        L = C.CDLL(LIB_PATH)
        L.az_last_error.restype = C.c_char_p
        for name, args in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        for name, args in _OPTIONAL.items():
            if hasattr(L, name):
                fn = getattr(L, name)
                fn.argtypes = args
                fn.restype = C.c_int
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise AzError(f"libazamd error {rc}: {lib().az_last_error().decode()}")


def exported_symbols():
    """Every symbol include/azamd.h declares (parsed from the header)."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "azamd.h")
    text = open(hdr).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(az_[a-z0-9_]+)\s*\(", text)))
