"""ctypes mirror of include/c8p_rx.h (the C ABI of libc8p_rx.so).

Field order and sizes must match the header exactly; tests/test_abi.py checks the sizes
against the values the library itself reports.
"""
import ctypes as C

import numpy as np

ABI_VERSION = 4
MEM_HOST, MEM_DEVICE, MEM_PINNED = 0, 1, 2
MEM_SC16, MEM_SC8 = 0x100, 0x200          # flags on the kind for RawChain.push*: int16 / int8 I/Q pairs
FMT_C32, FMT_SC16, FMT_SC8, FMT_F32 = 0, 1, 2, 3
PIECE_TO_HOST, PIECE_UNORDERED = 1, 2      # c8p_piece.rsv
F_L, F_HT, F_VHT, F_VHT_MU, F_VHT_CHAN = 0, 1, 2, 3, 20
CR_12, CR_23, CR_34, CR_56 = 0, 1, 2, 3
ST_OK, ST_REJ_VHT, ST_REJ_HT, ST_REJ_SHORT, ST_REJ_UNSUP = 0, 1, 2, 3, 4
BUF_TAGS, BUF_LLR, BUF_OUT, BUF_PDU, BUF_BITS, BUF_SBITS, BUF_MUCHAN, BUF_SIG_PKT, BUF_SIG_IQ, BUF_SIG_CHAN, \
    BUF_DET_EVT, BUF_DET_AC, BUF_DET_TRIG, BUF_DET_CONJ, BUF_TX_IQ, BUF_TX_DESC, BUF_TX_CODED, BUF_DECSTAT, BUF_DET_RAW = range(19)
K_LAYOUT, K_CLASSIFY, K_DEMOD, K_VITERBI, K_FINISH, K_SIGNAL, K_PRESISO, K_TRIGGER, K_SYNCEVT, K_TXBITS, K_TXSYM, K_COUNT = range(12)
K_NAMES = ["layout", "classify", "demod_sym", "viterbi_fwd", "finish", "signal", "presiso", "trigger", "syncevt", "txbits", "txsym"]
TAB_DEINT_L, TAB_DEINT_NL, TAB_DEINT_NL2, TAB_SIG_DEMAP_NL, TAB_DEINT_SIGB, TAB_SV_OUT, \
    TAB_PILOT_P, TAB_LTF_NL, TAB_LTF_NL_VHT22 = range(9)
DET_HIST = 80
TX_NO_WINDOW = 1
SIG_NEED_MORE, SIG_SKIPPED, SIG_PENDING = 1, 2, 3       # c8p_pkt_in.rsv[0] as c8p_signal writes it
NDP_TOTAL = 1024
NDP_BLOB_BYTES = 1027


class PktIn(C.Structure):
    _fields_ = [("iq_off", C.c_uint64 * 2), ("n_items", C.c_uint32), ("chan_off", C.c_uint32),
                ("lsig_mcs", C.c_int32), ("lsig_len", C.c_int32), ("lsig_nsamp", C.c_int32),
                ("seq", C.c_int32), ("cfo", C.c_float), ("snr", C.c_float), ("rssi", C.c_float),
                ("n_ant", C.c_uint8), ("mupos", C.c_uint8), ("rsv", C.c_uint8 * 2)]


class PktTags(C.Structure):
    _fields_ = [("status", C.c_int32), ("format", C.c_int32), ("mcs", C.c_int32), ("len", C.c_int32),
                ("cr", C.c_int32), ("ampdu", C.c_int32), ("trellis", C.c_int32), ("total", C.c_int32),
                ("nsym", C.c_int32), ("nss", C.c_int32), ("nltf", C.c_int32), ("ncbps", C.c_int32),
                ("ncbpss", C.c_int32), ("ndbps", C.c_int32), ("nbpscs", C.c_int32), ("nsymsamp", C.c_int32),
                ("sumu", C.c_int32), ("is_ndp", C.c_int32), ("data_off", C.c_int32), ("seq", C.c_int32),
                ("cfo", C.c_float), ("snr", C.c_float), ("rssi", C.c_float),
                ("sssnr0", C.c_float), ("sssnr1", C.c_float), ("rsv", C.c_uint32),
                ("llr_off", C.c_uint64)]


class PktOut(C.Structure):
    _fields_ = [("n_pdu", C.c_int32), ("n_crc_fail", C.c_int32), ("pdu_bytes", C.c_uint32),
                ("rsv", C.c_uint32), ("pdu_off", C.c_uint64), ("bits_off", C.c_uint64)]


class Cfg(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("profile", C.c_int32),
                ("keep_bits", C.c_int32), ("stream", C.c_void_p), ("keep_llr", C.c_int32)]


class BatchIn(C.Structure):
    _fields_ = [("n_pkt", C.c_int32), ("mem", C.c_int32), ("pkts", C.c_void_p), ("iq", C.c_void_p),
                ("n_iq", C.c_uint64), ("chan", C.c_void_p), ("n_chan", C.c_uint64)]


class SigIn(C.Structure):
    _fields_ = [("raw_off", C.c_uint64 * 2), ("n_avail", C.c_uint32), ("cfo_rad", C.c_float), ("snr", C.c_float),
                ("rssi", C.c_float), ("n_ant", C.c_uint8), ("rsv", C.c_uint8 * 3), ("stream", C.c_uint32)]


class SignalIn(C.Structure):
    _fields_ = [("n_evt", C.c_int32), ("mem", C.c_int32), ("evts", C.c_void_p), ("raw", C.c_void_p), ("n_raw", C.c_uint64),
                ("consume", C.c_int32), ("n_streams", C.c_int32), ("pkt_host", C.c_void_p), ("scan_from", C.c_uint64), ("scan_next", C.c_void_p),
                ("scan_from_v", C.c_void_p)]


class SyncEvt(C.Structure):
    _fields_ = [("index", C.c_uint64), ("rad", C.c_float), ("snr", C.c_float), ("rssi", C.c_float), ("trig", C.c_uint32)]


class DetState(C.Structure):
    _fields_ = [("run_start", C.c_int64), ("prev_start", C.c_int64), ("sync_pos", C.c_int64), ("in_run", C.c_int32),
                ("run_fired", C.c_int32), ("have_prev", C.c_int32), ("run_max", C.c_float), ("conj_avg", C.c_float * 2)]


class DetectIn(C.Structure):
    _fields_ = [("mem", C.c_int32), ("keep", C.c_int32), ("raw", C.c_void_p), ("n", C.c_uint64), ("n_hist", C.c_uint64),
                ("state", C.c_void_p), ("state_out", C.c_void_p), ("commit", C.c_void_p), ("evt_host", C.c_void_p), ("evt_cap", C.c_uint64)]


class DetStream(C.Structure):
    _fields_ = [("off", C.c_uint64), ("n", C.c_uint64), ("n_hist", C.c_uint64), ("commit", C.c_uint64), ("trig_off", C.c_uint64),
                ("n_evt", C.c_uint32), ("has_state", C.c_int32), ("state", DetState)]


class DetectMultiIn(C.Structure):
    _fields_ = [("mem", C.c_int32), ("n_streams", C.c_int32), ("raw", C.c_void_p), ("n_raw", C.c_uint64), ("streams", C.c_void_p),
                ("evt_host", C.c_void_p), ("evt_cap", C.c_uint64)]


class ReserveIn(C.Structure):
    _fields_ = [("rx_packets", C.c_int32), ("rx_ant", C.c_int32), ("rx_items", C.c_uint32), ("rx_host", C.c_int32),
                ("sig_events", C.c_int32), ("sig_host", C.c_int32), ("sig_raw", C.c_uint64), ("sig_items", C.c_uint64),
                ("det_items", C.c_uint64), ("det_host", C.c_int32), ("det_streams", C.c_int32), ("det_events", C.c_int32),
                ("det_keep", C.c_int32)]


class Span(C.Structure):
    _fields_ = [("dst_off", C.c_uint64), ("src_off", C.c_uint64), ("n", C.c_uint64)]


class Piece(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst_off", C.c_uint64), ("n", C.c_uint64), ("trusted", C.c_uint32), ("fmt", C.c_uint32),
                ("scale", C.c_float), ("rsv", C.c_uint32)]


class FetchPiece(C.Structure):
    _fields_ = [("which", C.c_int32), ("rsv", C.c_uint32), ("offset", C.c_uint64), ("bytes", C.c_uint64), ("dst", C.c_void_p)]


CNT_ARENA_GROWTHS, CNT_ARENA_BYTES, CNT_HOST_BYTES, CNT_DET_GIVEUPS, CNT_DET_TICKET_SCANS, CNT_SLOTS, CNT_SUBMITS, CNT_DET_TILE = range(8)


class TxPkt(C.Structure):
    _fields_ = [("psdu_off", C.c_uint64), ("len", C.c_uint32), ("format", C.c_uint8), ("mcs", C.c_uint8), ("scr_seed", C.c_uint8),
                ("flags", C.c_uint8), ("mult", C.c_float), ("gap", C.c_uint32), ("nss", C.c_uint8), ("rsv", C.c_uint8 * 3)]


class TxMu(C.Structure):
    _fields_ = [("psdu_off", C.c_uint64), ("len", C.c_uint32), ("mcs", C.c_uint8), ("group_id", C.c_uint8), ("rsv", C.c_uint8 * 2),
                ("q_off", C.c_uint64)]


class TxDesc(C.Structure):
    _fields_ = [("iq_off", C.c_uint64), ("coded_off", C.c_uint64), ("n_samp", C.c_uint32), ("n_coded", C.c_uint32),
                ("coded_off1", C.c_uint64), ("n_coded1", C.c_uint32), ("rsv", C.c_uint32)]


class TxIn(C.Structure):
    _fields_ = [("n_pkt", C.c_int32), ("mem", C.c_int32), ("pkts", C.c_void_p), ("psdu", C.c_void_p), ("n_psdu", C.c_uint64),
                ("tail_gap", C.c_uint32), ("n_mu", C.c_uint32), ("mu", C.c_void_p), ("bfq", C.c_void_p), ("n_bfq", C.c_uint64)]


class TxPlan(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("nb", "cr", "ndbps", "ncbps", "nsd", "nsym", "psdu_len", "npad_eof", "npad_octet",
                                         "legacy_len", "n_pre", "n_samp")]


class DecodeIn(C.Structure):
    _fields_ = [("n_pkt", C.c_int32), ("mem", C.c_int32), ("tags", C.c_void_p), ("llr", C.c_void_p),
                ("n_llr", C.c_uint64), ("mu2x1chan", C.c_void_p)]


def _np_dtype(struct):
    """numpy structured dtype with the same layout as a ctypes.Structure."""
    names, formats, offsets = [], [], []
    for name, ctype in struct._fields_:
        names.append(name)
        offsets.append(getattr(struct, name).offset)
        if hasattr(ctype, "_length_"):
            formats.append((np.dtype(ctype._type_), (ctype._length_,)))
        else:
            formats.append(np.dtype(ctype))
    return np.dtype({"names": names, "formats": formats, "offsets": offsets,
                     "itemsize": C.sizeof(struct)})


PKT_IN_DT = _np_dtype(PktIn)
PKT_TAGS_DT = _np_dtype(PktTags)
PKT_OUT_DT = _np_dtype(PktOut)
SIG_IN_DT = _np_dtype(SigIn)
SYNC_EVT_DT = _np_dtype(SyncEvt)
TX_PKT_DT = _np_dtype(TxPkt)
TX_DESC_DT = _np_dtype(TxDesc)
TX_MU_DT = _np_dtype(TxMu)
SPAN_DT = _np_dtype(Span)

TAG_KEYS = ("format", "mcs", "len", "cr", "ampdu", "trellis", "total")


def split_pdus(area: bytes, n_pdu: int):
    """PDU area = n_pdu records of {u16 len LE, len bytes}."""
    out, o = [], 0
    for _ in range(n_pdu):
        n = area[o] | (area[o + 1] << 8)
        out.append(bytes(area[o + 2:o + 2 + n]))
        o += 2 + n
    return out
