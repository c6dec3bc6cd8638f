// c8p_pinned_buffer.h — the buffer type the blocks of this module name for their stream ports: a GNU Radio 3.10 CUSTOM BUFFER whose
// memory is pinned host memory of the engine (c8p_host_alloc = hipHostMalloc), so that
//   * the block upstream of demod / demod2 (signal, signal2 — stock reference blocks) writes its items where the device can read
//     them: the front gathers the packets out of the scheduler's buffer with one kernel per general_work call (c8p_copy_gather)
//     instead of memcpy -> pinned staging -> hipMemcpyAsync;
//   * demod's LLR items are written into the demod -> decode edge by the device, and decode's front gathers them out of it again.
// flat_flowgraph::allocate_buffer takes an edge's buffer type from the io_signatures at its two ends (a custom type on either side
// wins over the default double-mapped one), so naming the type in OUR io_signatures is all a flowgraph needs: rx.grc loads unchanged.
//
// Modelled on gnuradio-runtime's own example of the mechanism, gr::host_buffer (gnuradio/host_buffer.h, lib/host_buffer.cc of 3.10):
// a gr::buffer_single_mapped whose do_allocate_buffer() provides the memory and whose blocked callbacks relocate unread items with
// the base class's *_logic helpers.  There is no device-side copy inside the buffer (post_work does nothing): the fronts move the items.
//
// A single-mapped buffer RELOCATES unread items when a reader or the writer is blocked at its end, and the fronts keep gathers under
// way on items they have cut and not consumed yet (batching_cfg::defer_consume: consuming them two calls later is worth 25 % of the
// fused chain's rate, 690-716 k against 546-559 k packets/s) and scatters on LLR items they have placed and not reported yet.  So both
// blocked callbacks first ANNOUNCE the move (c8p::buffer_move_scope: every front of the process waits for the copies it has under way
// on this buffer's memory — rare: once per trip round a 16 MB buffer — and the items a demod front has placed ahead of the write
// position are carried along) and only then let the base class move anything.  What a front
// has cut lives in its batch's device arena by then, and it counts in absolute items, not addresses: the move itself does not concern it.
//
// AUDIT LIST (continues the one in c8p_gr_adapter.h) — GNU Radio members used here, none of which the reference uses (it predates
// custom buffers); all as declared by gnuradio-runtime/include/gnuradio/{buffer.h,buffer_single_mapped.h,buffer_type.h,io_signature.h}
// of 3.10 as we read them (tests/gr_api_decls/ holds that reading):
//   gr::buffer_single_mapped(int nitems, size_t sizeof_item, uint64_t downstream_lcm_nitems, uint32_t downstream_max_out_mult,
//                            block_sptr link, block_sptr buf_owner)  [protected ctor]
//   gr::buffer_single_mapped::do_allocate_buffer(size_t final_nitems, size_t sizeof_item)   [pure virtual, protected]
//   gr::buffer::write_pointer(), gr::buffer::_read_pointer(unsigned read_index)            [pure virtual]
//   gr::buffer::post_work(int nitems)                                                       [pure virtual]
//   gr::buffer_single_mapped::input_blocked_callback(int items_required, int items_avail, unsigned read_index)   [pure virtual]
//   gr::buffer_single_mapped::output_blocked_callback(int output_multiple, bool force)                           [pure virtual]
//   gr::buffer_single_mapped::input_blocked_callback_logic(..., char* buffer_ptr, mem_func_t memcpy_func, mem_func_t memmove_func)
//   gr::buffer_single_mapped::output_blocked_callback_logic(int output_multiple, bool force, char* buffer_ptr, mem_func_t memmove_func)
//   gr::buffer::d_base, d_bufsize, d_sizeof_item, d_write_index, d_readers / gr::buffer_reader::d_read_index  [protected data]
//   gr::buffer::allocate_buffer(int nitems)                                                 [called by the factory]
//   gr::buffer_type (= const gr::buffer_type_base&), MAKE_CUSTOM_BUFFER_TYPE(NAME, factory) [gnuradio/buffer_type.h]
//   gr::io_signature::make(int min, int max, int sizeof_item, gr::buffer_type)              [gnuradio/io_signature.h]
//   gr::io_signature::stream_buffer_type(size_t index)                                      [the adapter's input_is_pinned()]
#ifndef INCLUDED_IEEE80211_C8P_PINNED_BUFFER_H
#define INCLUDED_IEEE80211_C8P_PINNED_BUFFER_H

#include <gnuradio/buffer_single_mapped.h>
#include <gnuradio/buffer_type.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#include "c8p_blocks.h"
#include "c8p_rx.h"

#ifndef C8P_PINNED_BUFFER_MIN_BYTES
#define C8P_PINNED_BUFFER_MIN_BYTES (16u << 20)  // 16 MB an edge: a general_work call then sees up to half of it (bench.py: "custom_buffers").  A
                                                  // front keeps two gathers of a third of its window under way (~45 us each from launch to done) while
                                                  // it cuts the next third: with 4 MB of samples a call that third is cut faster than a gather takes
                                                  // and the thread waits 0.35 us per packet (615 k packets/s fused); with 8 MB a call 0.26 (690 k)
#endif

namespace gr {
namespace ieee80211 {

class c8p_pinned_buffer : public gr::buffer_single_mapped
{
public:
  static gr::buffer_type type;

  ~c8p_pinned_buffer() override
  {
    if (!d_pinned) return;
    { std::lock_guard<std::mutex> g(reg_mu()); auto& r = reg(); r.erase(std::remove_if(r.begin(), r.end(), [this](const range& x) { return x.lo == (uintptr_t)d_pinned; }), r.end()); }
    c8p_host_free(d_pinned);
  }

  // is `p` inside a buffer this type has allocated?  (the adapter checks the pointers general_work is handed before a front lets the
  // device read or write them: an io_signature that names the type proves nothing about what the scheduler did with it)
  static bool owns(const void* p)
  {
    std::lock_guard<std::mutex> g(reg_mu());
    for (const range& x : reg()) if ((uintptr_t)p >= x.lo && (uintptr_t)p < x.hi) return true;
    return false;
  }

  // nothing to do after the writer's work(): the memory the writer wrote IS what the device reads
  void post_work(int /*nitems*/) override {}

  bool do_allocate_buffer(size_t final_nitems, size_t sizeof_item) override
  {
    d_pinned = static_cast<char*>(c8p_host_alloc((uint64_t)final_nitems * sizeof_item));
    if (!d_pinned) return false;
    d_base = d_pinned;
    { std::lock_guard<std::mutex> g(reg_mu()); reg().push_back(range{(uintptr_t)d_pinned, (uintptr_t)d_pinned + final_nitems * sizeof_item}); }
    return true;
  }

  void* write_pointer() override { return &d_base[d_write_index * d_sizeof_item]; }
  const void* _read_pointer(unsigned int read_index) override { return &d_base[read_index * d_sizeof_item]; }

  // Both callbacks may change d_write_index (input: += the blocked reader's gap; output: back by the smallest read index).  A demod
  // front in plain mode has LLR items placed AHEAD of d_write_index (the device writes them before produce() reports them): the scope
  // waits for them, saves them, and puts them back at the new write position (c8p_blocks.h, buffer_move_scope) — the base class moves
  // only what has been produced.  A front that still finds its output space somewhere else throws instead of reporting garbage.
  bool input_blocked_callback(int items_required, int items_avail, unsigned read_index) override
  {
    c8p::buffer_move_scope mv(d_base, (size_t)d_bufsize * d_sizeof_item);
    if (!mv.ok()) return false;                     // a front is busy on this memory and did not let go in time: not now
    const bool moved = input_blocked_callback_logic(items_required, items_avail, read_index, d_base, mem_copy, mem_move);
    mv.moved_to(write_pointer());
    return moved;
  }
  bool output_blocked_callback(int output_multiple, bool force) override
  {
    c8p::buffer_move_scope mv(d_base, (size_t)d_bufsize * d_sizeof_item);
    if (!mv.ok()) return false;
    const bool moved = output_blocked_callback_logic(output_multiple, force, d_base, mem_move);
    mv.moved_to(write_pointer());
    return moved;
  }

  // the factory gr::buffer_type carries (gnuradio/buffer_type.h: factory_func_ptr)
  static gr::buffer_sptr make_buffer(int nitems, size_t sizeof_item, uint64_t downstream_lcm_nitems, uint32_t downstream_max_out_mult,
                                     gr::block_sptr link, gr::block_sptr buf_owner)
  {
    // the scheduler asks for its default (64 KB); an edge of this module holds megabytes (see the header comment)
    const size_t want = ((size_t)C8P_PINNED_BUFFER_MIN_BYTES + sizeof_item - 1) / sizeof_item;
    if ((size_t)nitems < want) nitems = (int)want;
    c8p_pinned_buffer* b = new c8p_pinned_buffer(nitems, sizeof_item, downstream_lcm_nitems, downstream_max_out_mult, link, buf_owner);
    gr::buffer_sptr p(b);
    b->allocate_buffer(nitems);
    return p;
  }

private:
  c8p_pinned_buffer(int nitems, size_t sizeof_item, uint64_t downstream_lcm_nitems, uint32_t downstream_max_out_mult, gr::block_sptr link,
                    gr::block_sptr buf_owner)
      : gr::buffer_single_mapped(nitems, sizeof_item, downstream_lcm_nitems, downstream_max_out_mult, link, buf_owner)
  {
  }
  struct range { uintptr_t lo, hi; };
  static std::vector<range>& reg() { static std::vector<range> r; return r; }
  static std::mutex& reg_mu() { static std::mutex m; return m; }
  static void* mem_copy(void* d, const void* s, std::size_t n) { return std::memcpy(d, s, n); }
  static void* mem_move(void* d, const void* s, std::size_t n) { return std::memmove(d, s, n); }
  char* d_pinned = nullptr;
};

MAKE_CUSTOM_BUFFER_TYPE(C8P_PINNED, &c8p_pinned_buffer::make_buffer)
// (one definition per module: lib/demod_impl.cc defines  gr::buffer_type c8p_pinned_buffer::type(buftype_C8P_PINNED{});)

} // namespace ieee80211
} // namespace gr

#endif
