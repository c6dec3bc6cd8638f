// c8p_kernels.h — gfx950 kernels of the 802.11 OFDM RX hot path.
//
//   (arena layout is computed on the host from the descriptors and uploaded with them)
//   k_classify    wave / packet   format detection, SIG decode, channel estimate, SIG-B  (demod_impl.cc:106-219)
//   k_demod_sym   wave / symbol   FFT -> equalise -> pilot phase -> LLR -> de-interleave  (demod_impl.cc:279-314)
//   k_rx_1ss_lds / k_rx_2ss_lds   wave / packet   c8p_rx: symbol demodulation + lean forward pass, LLRs through LDS
//   k_viterbi_lean wave / packet  depuncture + 64-state ACS (c8p_viterbi_lean.h); viterbi_fwd_exact: the clamp-exact recurrence, inside k_finish
//   k_finish      wave / packet   traceback, descramble, A-MPDU walk, CRC-32, PDU records (decode_impl.cc:282-520)
//
// Every kernel is a single-wave (64-lane) program per unit of work: lane = FFT bin in the demod kernels,
// lane = trellis state in the decoder.  All reference citations are relative to /root/reference/lib/.
#pragma once
#include "c8p_dev.h"
#include "c8p_viterbi_lean.h"
#include "c8p_k_parse.h"       // SIG checks and parsers
#include "c8p_k_signal.h"      // k_signal
#include "c8p_k_classify.h"    // k_classify
#include "c8p_k_demod.h"       // demod_sym_1ss / demod_sym_2ss, k_demod_sym
#include "c8p_k_decode.h"      // k_llr_check, k_viterbi_lean, k_viterbi_lean4, viterbi_fwd_exact, redemod_packet
#include "c8p_k_rx.h"          // k_rx_1ss, k_rx_1ss_lds, k_rx_2ss_lds, k_viterbi_lean_lds
#include "c8p_k_finish.h"      // k_finish
#ifdef C8P_EXPERIMENTS
#include "c8p_k_exp.h"         // k_viterbi_lean2_lds (two packets per wave; measured and not adopted)
#endif
