"""ctypes binding of libcropsr_hip.so (C ABI: include/cropsr_hip.h).

The library is the only compute path of this package: if it is missing, or no
HIP device can be opened, the callers raise -- nothing falls back to the CPU.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CROPSR_HIP_LIB: load another build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get("CROPSR_HIP_LIB") or os.path.join(_HERE, "libcropsr_hip.so")

u8p = ctypes.POINTER(ctypes.c_uint8)
u32p = ctypes.POINTER(ctypes.c_uint32)
u64p = ctypes.POINTER(ctypes.c_uint64)
f64p = ctypes.POINTER(ctypes.c_double)


class RowSegment(ctypes.Structure):
    """crp_row_segment (include/cropsr_hip.h): consecutive rows of one contig with their columns, as crp_write_segments takes them."""
    _fields_ = [("contig_text", ctypes.c_void_p), ("contig_len", ctypes.c_uint64), ("chrom", ctypes.c_void_p), ("chrom_len", ctypes.c_uint64),
                ("pos", ctypes.c_void_p), ("minus", ctypes.c_void_p), ("score", ctypes.c_void_p), ("ids", ctypes.c_void_p),
                ("n_rows", ctypes.c_uint64), ("feat_blob", ctypes.c_void_p), ("feat_off", ctypes.c_void_p), ("feat_idx", ctypes.c_void_p),
                ("offtarget", ctypes.c_void_p)]

voidpp = ctypes.POINTER(ctypes.c_void_p)

# every symbol include/cropsr_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "crp_abi_version": (ctypes.c_int, []),
    "crp_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "crp_init": (ctypes.c_int, [ctypes.c_int, voidpp]),
    "crp_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "crp_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "crp_device_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int,
                                       ctypes.POINTER(ctypes.c_int), u64p]),
    "crp_arena_words_for": (ctypes.c_uint64, [ctypes.c_uint64]),
    "crp_arena_words_total": (ctypes.c_uint64, [ctypes.c_uint64]),
    "crp_arena_max_words": (ctypes.c_uint64, []),
    "crp_pack_ascii": (ctypes.c_int, [u8p, ctypes.c_uint64, u64p, u64p, u64p, u64p, ctypes.c_int]),
    "crp_arena_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, voidpp]),
    "crp_arena_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "crp_arena_add_contig_ascii": (ctypes.c_int, [ctypes.c_void_p, u8p, ctypes.c_uint64, u64p]),
    "crp_arena_add_contigs_ascii": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), u64p, ctypes.c_uint64, u64p]),
    "crp_arena_add_contig_packed": (ctypes.c_int, [ctypes.c_void_p, u64p, u64p, u64p, u64p,
                                                   ctypes.c_uint64, u64p]),
    "crp_arena_seal": (ctypes.c_int, [ctypes.c_void_p]),
    "crp_arena_tiles": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), u64p, u64p]),
    "crp_arena_stats": (ctypes.c_int, [ctypes.c_void_p, u64p, u64p, u64p]),
    "crp_arena_composition": (ctypes.c_int, [ctypes.c_void_p, u64p, u64p]),
    "crp_scan_score": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, u64p, u64p]),
    "crp_fetch_hits": (ctypes.c_int, [ctypes.c_void_p, u32p, f64p, f64p, u32p, f64p, f64p]),
    "crp_hits_counts": (ctypes.c_int, [ctypes.c_void_p, u64p, u64p]),
    "crp_hits_device": (ctypes.c_int, [ctypes.c_void_p, voidpp, voidpp, voidpp, voidpp]),
    "crp_score_30mers": (ctypes.c_int, [ctypes.c_void_p, u8p, ctypes.c_uint64, ctypes.c_int, f64p, f64p]),
    "crp_format_rows": (ctypes.c_int, [u8p, ctypes.c_uint64, u8p, ctypes.c_uint64, ctypes.c_int, u32p, u8p, f64p,
                                       u8p, ctypes.c_uint64, u8p, ctypes.c_uint64, u64p, ctypes.c_int]),
    "crp_write_rows": (ctypes.c_int, [ctypes.c_int, u8p, ctypes.c_uint64, u8p, ctypes.c_uint64, ctypes.c_int, u32p, u8p,
                                      f64p, u8p, ctypes.c_uint64, u64p, ctypes.c_int]),
    "crp_legacy_ids": (ctypes.c_int, [u32p, ctypes.POINTER(ctypes.c_int32), u8p, ctypes.c_uint64, ctypes.c_int]),
    "crp_fasta_table": (ctypes.c_int, [u8p, ctypes.c_uint64, u8p, ctypes.c_uint64, u64p, ctypes.c_uint64, u64p, u64p,
                                       ctypes.POINTER(ctypes.c_int), ctypes.c_int]),
    "crp_write_rows_ex": (ctypes.c_int, [ctypes.c_int, u8p, ctypes.c_uint64, u8p, ctypes.c_uint64, ctypes.c_int, u32p, u8p,
                                         f64p, u8p, ctypes.c_uint64, u8p, u64p, u32p, u32p, u64p, ctypes.c_int]),
    "crp_write_segments": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, u64p, ctypes.c_int]),
    "crp_comm_unique_id": (ctypes.c_int, [u8p]),
    "crp_comm_init": (ctypes.c_int, [ctypes.c_void_p, u8p, ctypes.c_int, ctypes.c_int]),
    "crp_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "crp_comm_barrier": (ctypes.c_int, [ctypes.c_void_p]),
    "crp_comm_allreduce_f64": (ctypes.c_int, [ctypes.c_void_p, f64p, ctypes.c_int, ctypes.c_int]),
    "crp_gather_hits": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, u64p]),
    "crp_gathered_fetch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, u32p, f64p, u32p, u32p, f64p, u32p]),
    "crp_gathered_fetch_features": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, u32p, u32p]),
    "crp_scan_stream": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_uint64, u32p, f64p, ctypes.c_uint64, u32p, f64p, ctypes.c_uint64, u64p, u64p, u64p, f64p]),
    "crp_scan_stream_prepare": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64]),
    "crp_host_alloc": (ctypes.c_int, [ctypes.c_uint64, voidpp]),
    "crp_host_free": (ctypes.c_int, [ctypes.c_void_p]),
    "crp_plan_shares": (ctypes.c_int, [u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, u64p, ctypes.c_uint64, u64p]),
    "crp_node_init": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), voidpp]),
    "crp_node_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "crp_node_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "crp_node_size": (ctypes.c_int, [ctypes.c_void_p]),
    "crp_node_ctx": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int]),
    "crp_node_arena": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int]),
    "crp_node_arenas": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "crp_node_arena_at": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "crp_node_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64]),
    "crp_node_transport_note": (ctypes.c_char_p, [ctypes.c_void_p]),
    "crp_node_comm_stuck": (ctypes.c_int, []),
    "crp_node_load": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), u64p, ctypes.c_uint64]),
    "crp_node_plan": (ctypes.c_int, [ctypes.c_void_p, u64p, ctypes.c_uint64, u64p]),
    "crp_node_scan_score": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, u64p, u64p]),
    "crp_node_offtarget": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, u64p]),
    "crp_node_annotate": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, u64p, ctypes.c_int]),
    "crp_node_fetch_offtarget": (ctypes.c_int, [ctypes.c_void_p, u32p, u32p]),
    "crp_node_fetch_features": (ctypes.c_int, [ctypes.c_void_p, u32p, u32p]),
    "crp_node_gather": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "crp_node_counts": (ctypes.c_int, [ctypes.c_void_p, u64p, u64p, u64p]),
    "crp_node_count_scored": (ctypes.c_int, [ctypes.c_void_p, u64p]),
    "crp_node_fetch": (ctypes.c_int, [ctypes.c_void_p, u32p, f64p, u32p, f64p]),
    "crp_node_tables_device": (ctypes.c_int, [ctypes.c_void_p, voidpp, voidpp, voidpp, voidpp]),
    "crp_node_gather_stats": (ctypes.c_int, [ctypes.c_void_p, f64p, f64p, u64p, ctypes.POINTER(ctypes.c_int)]),
    "crp_annotation_build": (ctypes.c_int, [u8p, ctypes.c_uint64, u8p, ctypes.c_uint64, voidpp]),
    "crp_annotation_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "crp_annotation_stats": (ctypes.c_int, [ctypes.c_void_p, u64p, u64p, u64p, u64p, u64p]),
    "crp_annotation_strings": (ctypes.c_int, [ctypes.c_void_p, u8p, u64p]),
    "crp_annotation_seqid": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, voidpp, u64p, voidpp, voidpp, u64p]),
    "crp_annotation_track": (ctypes.c_int, [ctypes.c_void_p, u64p, ctypes.c_uint64, ctypes.c_int, u32p, u32p, ctypes.c_uint64,
                                            u64p]),
    "crp_annotate_set_track": (ctypes.c_int, [ctypes.c_void_p, u32p, u32p, ctypes.c_uint64]),
    "crp_annotate_lookup": (ctypes.c_int, [ctypes.c_void_p, u32p, u32p]),
    "crp_offtarget_hist_get": (ctypes.c_int, [ctypes.c_void_p, u32p]),
    "crp_offtarget_hist_set": (ctypes.c_int, [ctypes.c_void_p, u32p]),
    "crp_offtarget_reset": (ctypes.c_int, [ctypes.c_void_p]),
    "crp_offtarget_add": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, u64p, ctypes.c_uint64, u64p]),
    "crp_offtarget_reduce": (ctypes.c_int, [ctypes.c_void_p]),
    "crp_offtarget_solve": (ctypes.c_int, [ctypes.c_void_p]),
    "crp_offtarget_counts": (ctypes.c_int, [ctypes.c_void_p, u32p, u32p]),
    "crp_offtarget_seeds": (ctypes.c_int, [ctypes.c_void_p, u32p, u32p]),
    "crp_configure": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64]),
    "crp_query": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]),
    "crp_build_id": (ctypes.c_char_p, []),
    "crp_profile_read_kind": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, f64p, u64p, ctypes.c_int]),
    "crp_count_scored": (ctypes.c_int, [ctypes.c_void_p, u64p]),
    "crp_profile_enable": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "crp_profile_read": (ctypes.c_int, [ctypes.c_void_p, f64p, u64p, ctypes.c_int]),
    "crp_synchronize": (ctypes.c_int, [ctypes.c_void_p]),
}

CRP_OK = 0
ORDER_BODY4, ORDER_TAIL2, ORDER_DOT1 = 0, 1, 2
OPT_TWO_PASS, OPT_CHAIN_TIMEOUT_US, OPT_TILE_GEOMETRY = 1, 2, 3
GEOMETRIES = {"auto": 0, "large": 1, "small": 2}
Q_CHAIN_TIMEOUTS, Q_TWO_PASS_ACTIVE, Q_COMM_WORLD, Q_COMM_RANK, Q_HBM_FREE, Q_HBM_TOTAL, Q_GATHER_BYTES = 1, 2, 3, 4, 5, 6, 7
KINDS = ("count", "tile_scan", "emit_score", "ot_seed", "ot_ball", "ot_lookup", "gatherv", "ot_reduce",
         "annotate")  # CRP_K_*
REDUCE_SUM, REDUCE_MAX = 0, 1
COMM_ID_BYTES = 128
GATHER_OFFTARGET, GATHER_PRE, GATHER_FEATURES, GATHER_POS16 = 1, 2, 4, 8
NODE_PEER_COPY, NODE_HOST_GATHER = 16, 32
NODE_OPT_ARENA_WORDS, NODE_OPT_COMM_INIT_TIMEOUT_MS, NODE_OPT_COLLECTIVE_TIMEOUT_MS = 1, 2, 3
TRANSPORTS = {0: "none (one device)", 1: "RCCL (in-library, one process)", 2: "device-to-device copies",
              3: "none: every device's rows over its own PCIe link to the host"}
HALO = 128
NO_FEATURE = 0xFFFFFFFF
SCAN_PRE, SCAN_SEEDS = 1, 2
OT_SEEDS = 1 << 24
OT_NOT_A_SITE, OT_NOT_OWNED = 0xFFFFFFFF, 0xFFFFFFFE
ABI_VERSION = 6
CRP_ERR_NO_DEVICE = -2
CRP_ERR_CAPACITY = -6
CRP_ERR_IO = -8
CRP_ERR_COMM = -9
CRP_ERR_NOMEM, CRP_ERR_STATE, CRP_ERR_PEER = -4, -5, -10
# what crp_gather_hits returns on EVERY rank together (agreed on before the exchange starts)
AGREED_GATHER_ERRORS = (CRP_ERR_NOMEM, CRP_ERR_STATE, CRP_ERR_PEER)

_lib = None


class CropsrHipError(RuntimeError):
    def __init__(self, status, what, detail=""):
        self.status = status
        msg = "%s failed: %s" % (what, lib().crp_strerror(status).decode())
        if detail:
            msg += " [%s]" % detail
        super().__init__(msg)


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C cropsr_amd/csrc`).  cropsr_amd has no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH, use_errno=True)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.crp_abi_version() != ABI_VERSION:
            raise ImportError("libcropsr_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(status, what, ctx=None):
    if status != CRP_OK:
        detail = lib().crp_last_error(ctx).decode() if ctx else ""
        raise CropsrHipError(status, what, detail)
