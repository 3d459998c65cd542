// mp_plan: a recorded sequence of C-ABI launches replayed by ONE call.
//
// The drop-in surface calls netG.filter once per frame from a stage thread (RTL/main.py:366-370,
// RTL/dataloader.py:1026-1053): on this library that is ~137 kernel launches, i.e. 137 foreign calls from an
// interpreter that re-acquires its global lock after every one of them while seven other stage threads want the
// same lock.  A plan is built once (the host records the calls of one real pass: same entry points, same
// argument structs) and replayed from C: one foreign call per frame, no interpreter in the loop.  Unlike a
// hipGraph it is a plain list of launches on the caller's stream (plus the plan's own side streams where the
// recorded pass used side streams, joined by events), so it composes with everything else on that stream and
// costs nothing to instantiate.  All buffers are the recorder's: the plan stores pointers only.
#include <vector>

#include "mp_internal.h"

struct mp_plan {
  mp_ctx *ctx = nullptr;
  struct Cmd {
    int kind = 0;
    int slot = 0;
    std::vector<unsigned char> blob;
    hipEvent_t event = nullptr;  // MP_PLAN_WAIT: recorded on the signaller, waited for by the waiter
  };
  std::vector<Cmd> cmds;
  std::vector<hipStream_t> side;  // slot k > 0 runs on side[k - 1]; slot 0 on the stream passed to mp_plan_run
};

namespace {

template <class T>
const T *blob_as(const mp_plan::Cmd &c) {
  return c.blob.size() == sizeof(T) ? reinterpret_cast<const T *>(c.blob.data()) : nullptr;
}

size_t blob_size(int kind) {
  switch (kind) {
    case MP_PLAN_CONVK: return sizeof(mp_convk_args);
    case MP_PLAN_GN_APPLY: return sizeof(mp_plan_gn_apply_args);
    case MP_PLAN_CONV3X3: return sizeof(mp_conv3x3_args);
    case MP_PLAN_CONV1X1: return sizeof(mp_conv1x1_args);
    case MP_PLAN_AVGPOOL2: return sizeof(mp_plan_pool_args);
    case MP_PLAN_UPSAMPLE2X: return sizeof(mp_plan_upsample_args);
    case MP_PLAN_MEMSET: return sizeof(mp_plan_memset_args);
    case MP_PLAN_WAIT: return sizeof(mp_plan_wait_args);
    default: return 0;
  }
}

}  // namespace

extern "C" {

int mp_plan_create(mp_ctx *ctx, int n_side_streams, mp_plan **out) {
  if (!ctx || !out || n_side_streams < 0 || n_side_streams > 8)
    return ctx ? mp::fail(ctx, MP_ERR_ARG, "mp_plan_create: bad argument") : MP_ERR_ARG;
  mp_plan *p = new mp_plan();
  p->ctx = ctx;
  int prev = 0;
  hipGetDevice(&prev);
  hipSetDevice(ctx->device);
  for (int i = 0; i < n_side_streams; ++i) {
    hipStream_t s = nullptr;
    // MONOPORT_PLAN_SIDE_PRIORITY (measurement switch): HIP priority of the side streams (lower = served first)
    static const char *prio_env = getenv("MONOPORT_PLAN_SIDE_PRIORITY");
    const hipError_t e_create = prio_env ? hipStreamCreateWithPriority(&s, hipStreamNonBlocking, atoi(prio_env))
                                         : hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e_create != hipSuccess) {
      for (hipStream_t t : p->side) hipStreamDestroy(t);
      delete p;
      hipSetDevice(prev);
      return mp::fail(ctx, MP_ERR_HIP, "mp_plan_create: hipStreamCreate failed");
    }
    p->side.push_back(s);
  }
  hipSetDevice(prev);
  *out = p;
  return MP_OK;
}

int mp_plan_add(mp_plan *plan, int kind, const void *args, int64_t bytes, int stream_slot) {
  if (!plan) return MP_ERR_ARG;
  mp_ctx *ctx = plan->ctx;
  const size_t want = blob_size(kind);
  if (!args || want == 0 || (size_t)bytes != want)
    return mp::fail(ctx, MP_ERR_ARG, "mp_plan_add: kind %d takes %zu bytes of arguments, got %lld", kind, want,
                    (long long)bytes);
  if (stream_slot < 0 || stream_slot > (int)plan->side.size())
    return mp::fail(ctx, MP_ERR_ARG, "mp_plan_add: stream slot %d of %zu", stream_slot, plan->side.size() + 1);
  mp_plan::Cmd c;
  c.kind = kind;
  c.slot = stream_slot;
  c.blob.assign(static_cast<const unsigned char *>(args), static_cast<const unsigned char *>(args) + bytes);
  if (kind == MP_PLAN_WAIT) {
    const mp_plan_wait_args *w = blob_as<mp_plan_wait_args>(c);
    const int n = (int)plan->side.size();
    if (w->waiter_slot < 0 || w->waiter_slot > n || w->signaller_slot < 0 || w->signaller_slot > n ||
        w->waiter_slot == w->signaller_slot)
      return mp::fail(ctx, MP_ERR_ARG, "mp_plan_add: wait %d <- %d", w->waiter_slot, w->signaller_slot);
    int prev = 0;
    hipGetDevice(&prev);
    hipSetDevice(ctx->device);
    const hipError_t e = hipEventCreateWithFlags(&c.event, hipEventDisableTiming);
    hipSetDevice(prev);
    if (e != hipSuccess) return mp::fail(ctx, MP_ERR_HIP, "mp_plan_add: hipEventCreate failed");
  }
  plan->cmds.push_back(std::move(c));
  return MP_OK;
}

int mp_plan_size(mp_plan *plan) { return plan ? (int)plan->cmds.size() : 0; }

int mp_plan_run(mp_plan *plan, mp_stream stream) {
  if (!plan) return MP_ERR_ARG;
  mp_ctx *ctx = plan->ctx;
  auto st = [&](int slot) { return slot == 0 ? (hipStream_t)stream : plan->side[slot - 1]; };
  int prev = 0;
  hipGetDevice(&prev);
  hipSetDevice(ctx->device);
  int rc = MP_OK;
  for (const mp_plan::Cmd &c : plan->cmds) {
    mp_stream s = (mp_stream)st(c.slot);
    switch (c.kind) {
      case MP_PLAN_CONVK: rc = mp_convk(ctx, blob_as<mp_convk_args>(c), s); break;
      case MP_PLAN_CONV3X3: rc = mp_conv3x3_ex(ctx, blob_as<mp_conv3x3_args>(c), s); break;
      case MP_PLAN_CONV1X1: rc = mp_conv1x1_ex(ctx, blob_as<mp_conv1x1_args>(c), s); break;
      case MP_PLAN_GN_APPLY: {
        const mp_plan_gn_apply_args *a = blob_as<mp_plan_gn_apply_args>(c);
        rc = mp_gn_apply(ctx, a->x, &a->gn, a->relu, a->n, a->c, a->hw, a->res, a->y, &a->fin, s);
        break;
      }
      case MP_PLAN_AVGPOOL2: {
        const mp_plan_pool_args *a = blob_as<mp_plan_pool_args>(c);
        rc = mp_avgpool2_gn(ctx, a->x, a->n, a->c, a->h, a->w, a->y, &a->fin, s);
        break;
      }
      case MP_PLAN_UPSAMPLE2X: {
        const mp_plan_upsample_args *a = blob_as<mp_plan_upsample_args>(c);
        rc = mp_upsample_bicubic2x_gn(ctx, a->x, a->n, a->c, a->h, a->w, a->add, a->y, &a->fin, s);
        break;
      }
      case MP_PLAN_MEMSET: {
        const mp_plan_memset_args *a = blob_as<mp_plan_memset_args>(c);
        if (hipMemsetAsync(a->ptr, a->value, (size_t)a->bytes, (hipStream_t)s) != hipSuccess)
          rc = mp::fail(ctx, MP_ERR_HIP, "mp_plan_run: hipMemsetAsync failed");
        break;
      }
      case MP_PLAN_WAIT: {
        const mp_plan_wait_args *a = blob_as<mp_plan_wait_args>(c);
        if (hipEventRecord(c.event, st(a->signaller_slot)) != hipSuccess ||
            hipStreamWaitEvent(st(a->waiter_slot), c.event, 0) != hipSuccess)
          rc = mp::fail(ctx, MP_ERR_HIP, "mp_plan_run: event record / wait failed");
        break;
      }
      default: rc = mp::fail(ctx, MP_ERR_STATE, "mp_plan_run: unknown command %d", c.kind);
    }
    if (rc != MP_OK) break;
  }
  if (rc != MP_OK && !plan->side.empty()) {
    // a command failed in the middle of the sequence: whatever the side streams were given so far must not be
    // left running unordered against the caller's stream (the caller may free or reuse the plan's buffers as soon
    // as ITS stream is done) -- join every side stream into slot 0 before reporting the error
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
      for (hipStream_t s : plan->side)
        if (hipEventRecord(ev, s) == hipSuccess) (void)hipStreamWaitEvent((hipStream_t)stream, ev, 0);
      (void)hipEventDestroy(ev);
    }
    (void)hipGetLastError();
  }
  hipSetDevice(prev);
  return rc;
}

void mp_plan_destroy(mp_plan *plan) {
  if (!plan) return;
  int prev = 0;
  hipGetDevice(&prev);
  hipSetDevice(plan->ctx->device);
  for (hipStream_t s : plan->side) {
    hipStreamSynchronize(s);
    hipStreamDestroy(s);
  }
  for (mp_plan::Cmd &c : plan->cmds)
    if (c.event) hipEventDestroy(c.event);
  hipSetDevice(prev);
  delete plan;
}

}  // extern "C"
