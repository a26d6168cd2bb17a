// s3d_sweep.h — C1: a loop-closure sweep sharded over the GPUs of ONE node, in one process (included by
// s3d_api.hip; entry points declared in include/slam3d_hip.h).
//
// What it replaces: the serial candidate loop of ScanSensor::linkToNeighbors (ScanSensor.cpp:170-202; also entered
// from the detached link thread, :209-210), which calls createConstraint once per candidate pair on the CPU.
// Here the candidate pairs are cut into contiguous blocks, one block per GPU (= rank); every rank has its own
// s3d_context and host thread, uploads only the clouds its block references (once: they stay resident for later
// sweeps) and runs ONE s3d_align_batch.  The pairs are independent, so there is no data-path collective; the
// only exchange is one all-gather of the 128-byte edge records (RCCL, ncclAllGather over xGMI) that leaves every
// edge in every GPU's HBM in pair order; the host result is read back from rank 0's gathered buffer.
//
// RCCL is bound at run time (dlopen of librccl.so.1: a process that already has torch's RCCL loaded gets that
// one).  RCCL admits one rank per device: when the device list names a device twice (two contexts on one GPU, the
// 1-GPU test configuration) the gather is done with device-to-device copies instead and s3d_sweep_collective()
// says "copy".  A pair's record does not depend on the shard it lands in (block_reduce_store_fixed), so the sweep
// returns the records of the single-context s3d_align_batch bit for bit for any number of ranks.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <functional>
#include <thread>

struct s3d_sweep_cloud {
  float* xyz = nullptr;               // PINNED host copy (stride floats per point): uploaded to a rank on first use
  int n = 0, stride = 3;              // by DMA straight from here (a pageable copy goes through a staging bounce)
  std::vector<s3d_cloud*> dev;        // per rank, nullptr until a pair of that rank's block references the cloud
};

namespace {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string* why) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)   // a copy that is already mapped (torch's) wins over loading a second one
      if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD))) break;
    for (int i = 0; !lib && i < 3; ++i) lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { *why = std::string("librccl not found: ") + dlerror(); return false; }
#define S3D_SYM(field, name) \
    field = reinterpret_cast<decltype(field)>(dlsym(lib, name)); \
    if (!field) { *why = std::string("librccl lacks ") + name; return false; }
    S3D_SYM(CommInitAll, "ncclCommInitAll") S3D_SYM(CommDestroy, "ncclCommDestroy") S3D_SYM(AllGather, "ncclAllGather")
    S3D_SYM(GroupStart, "ncclGroupStart") S3D_SYM(GroupEnd, "ncclGroupEnd") S3D_SYM(GetErrorString, "ncclGetErrorString")
#undef S3D_SYM
    return true;
  }
};

}  // namespace

// the host thread of a rank: started once with the sweep and kept between calls (a sweep of a few hundred short
// registrations should not pay thread creation and the first-touch of a new thread's HIP state every time)
struct RankWorker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, stop = false, done = true;
  void start() {
    th = std::thread([this] {
      std::unique_lock<std::mutex> lk(m);
      for (;;) {
        cv.wait(lk, [this] { return has_job || stop; });
        if (stop) return;
        std::function<void()> j = std::move(job);
        has_job = false;
        lk.unlock();
        try { j(); } catch (...) {}   // (the jobs report through their status slots; nothing may leave a thread's function)
        lk.lock();
        done = true;
        cv.notify_all();
      }
    });
  }
  void post(std::function<void()> j) {
    std::lock_guard<std::mutex> lk(m);
    job = std::move(j); has_job = true; done = false;
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [this] { return done; });
  }
  void shutdown() {
    { std::lock_guard<std::mutex> lk(m); stop = true; }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};

struct s3d_sweep {
  std::vector<std::unique_ptr<RankWorker>> workers;   // one per rank (none for a single rank: the caller's thread)
  std::vector<s3d_edge_record*> stage;                // pinned staging of a rank's records
  std::vector<size_t> stage_cap;
  std::vector<int> devices;           // rank -> HIP device
  std::vector<s3d_context*> ctx;      // one context (stream + workspace) per rank
  std::vector<hipStream_t> coll;      // the stream the collective of a rank runs on
  std::vector<void*> sendbuf, recvbuf;
  std::vector<size_t> send_cap, recv_cap;
  std::vector<ncclComm_t> comms;
  RcclApi rccl;
  bool use_rccl = false;
  std::string err, collective = "copy";
  std::mutex mtx;
  int R() const { return (int)devices.size(); }
};

extern "C" {

void s3d_sweep_shard_range(int n_pairs, int n_ranks, int rank, int* lo, int* hi) try {
  // contiguous blocks of ceil(n / ranks) pairs; the last ranks may be short or empty (== slam3d_amd/sweep.py)
  const int per = n_ranks > 0 ? (n_pairs + n_ranks - 1) / n_ranks : 0;
  const int l = std::min(rank * per, n_pairs);
  if (lo) *lo = l;
  if (hi) *hi = std::min(l + per, n_pairs);
} catch (...) {}   // (a destructor-like entry point has no status to return)

void s3d_sweep_destroy(s3d_sweep* sw) try {
  if (!sw) return;
  for (auto& w : sw->workers) if (w) w->shutdown();
  for (s3d_edge_record* p : sw->stage) if (p) (void)hipHostFree(p);
  for (int r = 0; r < sw->R(); ++r) {
    (void)hipSetDevice(sw->devices[r]);
    if (r < (int)sw->comms.size() && sw->comms[r]) (void)sw->rccl.CommDestroy(sw->comms[r]);
    if (r < (int)sw->coll.size() && sw->coll[r]) { (void)hipStreamSynchronize(sw->coll[r]); (void)hipStreamDestroy(sw->coll[r]); }
    if (r < (int)sw->sendbuf.size() && sw->sendbuf[r]) (void)hipFree(sw->sendbuf[r]);
    if (r < (int)sw->recvbuf.size() && sw->recvbuf[r]) (void)hipFree(sw->recvbuf[r]);
    if (r < (int)sw->ctx.size() && sw->ctx[r]) s3d_context_destroy(sw->ctx[r]);
  }
  delete sw;
} catch (...) {}   // (a destructor-like entry point has no status to return)

static int sweep_create(int n_devices, const int* devices, const uint32_t* cu_mask, int cu_words, s3d_sweep** out);
int s3d_sweep_create(int n_devices, const int* devices, s3d_sweep** out) try {
  return sweep_create(n_devices, devices, nullptr, 0, out);
} catch (...) { return fail_current(nullptr); }
int s3d_sweep_create_cu_mask(int n_devices, const int* devices, const uint32_t* cu_mask, int n_words, s3d_sweep** out) try {
  if (!cu_mask || n_words <= 0 || n_words > 32) return S3D_STATUS_INVALID_ARGUMENT;
  return sweep_create(n_devices, devices, cu_mask, n_words, out);
} catch (...) { return fail_current(nullptr); }
static int sweep_create(int n_devices, const int* devices, const uint32_t* cu_mask, int cu_words, s3d_sweep** out) {
  if (!out || n_devices < 0) return S3D_STATUS_INVALID_ARGUMENT;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return S3D_STATUS_BACKEND_ERROR;   // no CPU fallback
  s3d_sweep* sw = new s3d_sweep();
  try {
  if (n_devices == 0 || !devices) {
    const int n = n_devices == 0 ? count : n_devices;
    for (int d = 0; d < n; ++d) sw->devices.push_back(d % count);
  } else {
    sw->devices.assign(devices, devices + n_devices);
  }
  const int R = sw->R();
  bool distinct = true;
  for (int a = 0; a < R; ++a) {
    if (sw->devices[a] < 0 || sw->devices[a] >= count) { delete sw; return S3D_STATUS_INVALID_ARGUMENT; }
    for (int b = 0; b < a; ++b) distinct = distinct && sw->devices[a] != sw->devices[b];
  }
  sw->ctx.assign(R, nullptr); sw->coll.assign(R, nullptr);
  sw->sendbuf.assign(R, nullptr); sw->recvbuf.assign(R, nullptr);
  sw->send_cap.assign(R, 0); sw->recv_cap.assign(R, 0);
  sw->stage.assign(R, nullptr); sw->stage_cap.assign(R, 0);
  if (R > 1)
    for (int r = 0; r < R; ++r) { sw->workers.emplace_back(new RankWorker()); sw->workers.back()->start(); }
  for (int r = 0; r < R; ++r) {
    if ((cu_mask ? s3d_context_create_cu_mask(sw->devices[r], cu_mask, cu_words, &sw->ctx[r])
                 : s3d_context_create(sw->devices[r], nullptr, &sw->ctx[r])) != S3D_STATUS_OK ||
        hipSetDevice(sw->devices[r]) != hipSuccess ||
        hipStreamCreateWithFlags(&sw->coll[r], hipStreamNonBlocking) != hipSuccess) {
      s3d_sweep_destroy(sw);
      return S3D_STATUS_BACKEND_ERROR;
    }
  }
  // RCCL communicators: one rank per DISTINCT device (RCCL refuses a device named twice)
  if (distinct) {
    std::string why;
    if (sw->rccl.load(&why)) {
      sw->comms.assign(R, nullptr);
      const ncclResult_t rc = sw->rccl.CommInitAll(sw->comms.data(), R, sw->devices.data());
      if (rc == ncclSuccess) {
        sw->use_rccl = true;
        sw->collective = "rccl";
      } else {
        sw->err = std::string("ncclCommInitAll: ") + sw->rccl.GetErrorString(rc);
        sw->comms.clear();
      }
    } else {
      sw->err = why;
    }
    if (!sw->use_rccl && R > 1) {   // several GPUs without RCCL is a broken installation, not a mode to run in silently
      s3d_sweep_destroy(sw);
      return S3D_STATUS_BACKEND_ERROR;
    }
  }
  } catch (...) {   // (a host container that could not grow, a worker thread that could not start)
    s3d_sweep_destroy(sw);
    return fail_current(nullptr);
  }
  *out = sw;
  return S3D_STATUS_OK;
}

int s3d_sweep_ranks(const s3d_sweep* sw) { return sw ? sw->R() : 0; }
const char* s3d_sweep_collective(const s3d_sweep* sw) { return sw ? sw->collective.c_str() : ""; }
const char* s3d_sweep_last_error(const s3d_sweep* sw) { return sw ? sw->err.c_str() : "null sweep"; }
s3d_context* s3d_sweep_context(s3d_sweep* sw, int rank) { return (sw && rank >= 0 && rank < sw->R()) ? sw->ctx[rank] : nullptr; }

int s3d_sweep_cloud_create(s3d_sweep* sw, const float* xyz, int n, int stride, s3d_sweep_cloud** out) try {
  if (!sw || !out || n < 0 || stride < 3 || (n > 0 && !xyz)) return S3D_STATUS_INVALID_ARGUMENT;
  s3d_sweep_cloud* c = new s3d_sweep_cloud();
  c->n = n; c->stride = stride;
  if (n > 0) {
    // n * stride floats (the upload of stride-4 points copies whole 16-byte records), the caller's last point may
    // end after its third float
    const size_t count = (size_t)(n - 1) * stride + 3, full = (size_t)n * stride;
    if (hipHostMalloc((void**)&c->xyz, full * sizeof(float)) != hipSuccess) { delete c; return S3D_STATUS_BACKEND_ERROR; }
    std::memcpy(c->xyz, xyz, count * sizeof(float));
    for (size_t k = count; k < full; ++k) c->xyz[k] = 0.f;
  }
  c->dev.assign(sw->R(), nullptr);
  *out = c;
  return S3D_STATUS_OK;
} catch (...) { return fail_current(nullptr, sw ? &sw->err : nullptr); }

void s3d_sweep_cloud_release(s3d_sweep* sw, s3d_sweep_cloud* c) try {
  if (!c) return;
  if (sw) {
    std::lock_guard<std::mutex> lock(sw->mtx);
    for (int r = 0; r < sw->R() && r < (int)c->dev.size(); ++r)
      if (c->dev[r]) s3d_cloud_release(sw->ctx[r], c->dev[r]);
  }
  if (c->xyz) (void)hipHostFree(c->xyz);
  delete c;
} catch (...) {}   // (a destructor-like entry point has no status to return)

int s3d_align_batch_multi(s3d_sweep* sw, int n_pairs, s3d_sweep_cloud* const* sources, s3d_sweep_cloud* const* targets,
                          const double* guesses, const s3d_reg_params* params, const s3d_exec_options* opts,
                          s3d_edge_record* records) try {
  if (!sw || n_pairs < 0 || !params || (n_pairs > 0 && (!sources || !targets || !guesses || !records)))
    return S3D_STATUS_INVALID_ARGUMENT;
  for (int p = 0; p < n_pairs; ++p)
    if (!sources[p] || !targets[p]) return S3D_STATUS_INVALID_ARGUMENT;
  if (n_pairs == 0) return S3D_STATUS_OK;
  std::lock_guard<std::mutex> lock(sw->mtx);
  const int R = sw->R();
  const int per = (n_pairs + R - 1) / R;
  const size_t rec_bytes = sizeof(s3d_edge_record);
  std::vector<int> status(R, S3D_STATUS_OK);
  std::vector<std::string> errs(R);
  // ---- every rank: its block of the pair list on its own GPU, records staged in its send buffer
  auto work_body = [&](int r) {
    int lo, hi;
    s3d_sweep_shard_range(n_pairs, R, r, &lo, &hi);
    const int m = hi - lo;
    auto hipok = [&](hipError_t e, const char* what) {
      if (e == hipSuccess) return true;
      status[r] = S3D_STATUS_BACKEND_ERROR;
      errs[r] = std::string(what) + ": " + hipGetErrorString(e);
      return false;
    };
    if (!hipok(hipSetDevice(sw->devices[r]), "hipSetDevice")) return;
    if (sw->send_cap[r] < (size_t)per * rec_bytes) {
      if (sw->sendbuf[r]) (void)hipFree(sw->sendbuf[r]);
      sw->sendbuf[r] = nullptr; sw->send_cap[r] = 0;
      if (!hipok(hipMalloc(&sw->sendbuf[r], (size_t)per * rec_bytes), "hipMalloc(send)")) return;
      sw->send_cap[r] = (size_t)per * rec_bytes;
    }
    if (sw->recv_cap[r] < (size_t)per * R * rec_bytes) {
      if (sw->recvbuf[r]) (void)hipFree(sw->recvbuf[r]);
      sw->recvbuf[r] = nullptr; sw->recv_cap[r] = 0;
      if (!hipok(hipMalloc(&sw->recvbuf[r], (size_t)per * R * rec_bytes), "hipMalloc(recv)")) return;
      sw->recv_cap[r] = (size_t)per * R * rec_bytes;
    }
    if (sw->stage_cap[r] < (size_t)per) {
      if (sw->stage[r]) (void)hipHostFree(sw->stage[r]);
      sw->stage[r] = nullptr; sw->stage_cap[r] = 0;
      if (!hipok(hipHostMalloc((void**)&sw->stage[r], (size_t)per * rec_bytes), "hipHostMalloc(records)")) return;
      sw->stage_cap[r] = (size_t)per;
    }
    s3d_edge_record* local = sw->stage[r];
    std::memset(local, 0, (size_t)per * rec_bytes);                  // padding of a short block
    if (m > 0) {
      std::vector<s3d_cloud*> src((size_t)m), tgt((size_t)m);
      for (int p = 0; p < m; ++p) {
        s3d_sweep_cloud* pair_clouds[2] = {sources[lo + p], targets[lo + p]};
        for (s3d_sweep_cloud* c : pair_clouds) {
          if (!c->dev[r]) {   // first use on this GPU: upload (only this rank's thread touches dev[r])
            const int st = s3d_cloud_upload(sw->ctx[r], c->xyz, c->n, c->stride, &c->dev[r]);
            if (st != S3D_STATUS_OK) { status[r] = st; errs[r] = s3d_last_error(sw->ctx[r]); return; }
          }
        }
        src[p] = sources[lo + p]->dev[r];
        tgt[p] = targets[lo + p]->dev[r];
      }
      const int st = s3d_align_batch(sw->ctx[r], m, src.data(), tgt.data(), guesses + (size_t)lo * 16, params, opts,
                                     local, nullptr);
      if (st != S3D_STATUS_OK) {   // (unknown algorithm: the records carry the status, as in s3d_align_batch)
        status[r] = st;
        if (st == S3D_STATUS_BACKEND_ERROR) { errs[r] = s3d_last_error(sw->ctx[r]); return; }
      }
      if (!hipok(hipSetDevice(sw->devices[r]), "hipSetDevice")) return;
    }
    if (!hipok(hipMemcpyAsync(sw->sendbuf[r], local, (size_t)per * rec_bytes, hipMemcpyHostToDevice, sw->coll[r]),
               "hipMemcpyAsync(records)")) return;
    (void)hipok(hipStreamSynchronize(sw->coll[r]), "hipStreamSynchronize");   // (the next call reuses the staging)
  };
  auto work = [&](int r) {   // a rank's exception becomes its status (the worker thread must not be left by unwinding)
    try { work_body(r); } catch (...) { status[r] = fail_current(nullptr, &errs[r]); if (status[r] != S3D_STATUS_BACKEND_ERROR) status[r] = S3D_STATUS_BACKEND_ERROR; }
  };
  if (R == 1) {
    work(0);
  } else {
    for (int r = 0; r < R; ++r) sw->workers[(size_t)r]->post([&work, r] { work(r); });
    for (int r = 0; r < R; ++r) sw->workers[(size_t)r]->wait();
  }
  int worst = S3D_STATUS_OK;
  for (int r = 0; r < R; ++r) {
    if (status[r] == S3D_STATUS_BACKEND_ERROR) { sw->err = "rank " + std::to_string(r) + ": " + errs[r]; return S3D_STATUS_BACKEND_ERROR; }
    if (status[r] != S3D_STATUS_OK) worst = status[r];
  }
  // ---- the exchange step: all-gather of the records, every rank ends up with all of them in pair order
  const size_t count = (size_t)per * (rec_bytes / sizeof(double));
  if (sw->use_rccl) {
    ncclResult_t rc = sw->rccl.GroupStart();
    for (int r = 0; r < R && rc == ncclSuccess; ++r)
      rc = sw->rccl.AllGather(sw->sendbuf[r], sw->recvbuf[r], count, ncclDouble, sw->comms[r], sw->coll[r]);
    const ncclResult_t rc2 = sw->rccl.GroupEnd();
    if (rc == ncclSuccess) rc = rc2;
    if (rc != ncclSuccess) { sw->err = std::string("ncclAllGather: ") + sw->rccl.GetErrorString(rc); return S3D_STATUS_BACKEND_ERROR; }
  } else {
    for (int r = 0; r < R; ++r) {
      if (hipSetDevice(sw->devices[r]) != hipSuccess) return S3D_STATUS_BACKEND_ERROR;
      for (int q = 0; q < R; ++q)
        if (hipMemcpyAsync((char*)sw->recvbuf[r] + (size_t)q * per * rec_bytes, sw->sendbuf[q], (size_t)per * rec_bytes,
                           hipMemcpyDeviceToDevice, sw->coll[r]) != hipSuccess) {
          sw->err = "device-to-device gather failed";
          return S3D_STATUS_BACKEND_ERROR;
        }
    }
  }
  for (int r = 0; r < R; ++r) {
    if (hipSetDevice(sw->devices[r]) != hipSuccess || hipStreamSynchronize(sw->coll[r]) != hipSuccess) {
      sw->err = "collective stream failed on rank " + std::to_string(r);
      return S3D_STATUS_BACKEND_ERROR;
    }
  }
  // ---- host result: rank 0's gathered buffer, padding of short blocks dropped
  std::vector<s3d_edge_record> all((size_t)per * R);
  // (on the rank's collective stream, not the null stream: see copy_to_host in s3d_api.hip)
  if (hipSetDevice(sw->devices[0]) != hipSuccess ||
      hipMemcpyAsync(all.data(), sw->recvbuf[0], all.size() * rec_bytes, hipMemcpyDeviceToHost, sw->coll[0]) != hipSuccess ||
      hipStreamSynchronize(sw->coll[0]) != hipSuccess) {
    sw->err = "download of the gathered records failed";
    return S3D_STATUS_BACKEND_ERROR;
  }
  for (int r = 0; r < R; ++r) {
    int lo, hi;
    s3d_sweep_shard_range(n_pairs, R, r, &lo, &hi);
    if (hi > lo) std::memcpy(records + lo, all.data() + (size_t)r * per, (size_t)(hi - lo) * rec_bytes);
  }
  return worst;
} catch (...) { return fail_current(nullptr, sw ? &sw->err : nullptr); }

// the gathered records as rank `rank` holds them in HBM after the last s3d_align_batch_multi (tests: every rank
// must hold every edge): n_pairs records in pair order
int s3d_sweep_gathered_records(s3d_sweep* sw, int rank, int n_pairs, s3d_edge_record* records) try {
  if (!sw || rank < 0 || rank >= sw->R() || n_pairs < 0 || (n_pairs > 0 && !records)) return S3D_STATUS_INVALID_ARGUMENT;
  if (n_pairs == 0) return S3D_STATUS_OK;
  std::lock_guard<std::mutex> lock(sw->mtx);
  const int R = sw->R();
  const int per = (n_pairs + R - 1) / R;
  const size_t rec_bytes = sizeof(s3d_edge_record);
  if (sw->recv_cap[rank] < (size_t)per * R * rec_bytes) return S3D_STATUS_INVALID_ARGUMENT;
  std::vector<s3d_edge_record> all((size_t)per * R);
  if (hipSetDevice(sw->devices[rank]) != hipSuccess ||
      hipMemcpyAsync(all.data(), sw->recvbuf[rank], all.size() * rec_bytes, hipMemcpyDeviceToHost, sw->coll[rank]) != hipSuccess ||
      hipStreamSynchronize(sw->coll[rank]) != hipSuccess)
    return S3D_STATUS_BACKEND_ERROR;
  for (int r = 0; r < R; ++r) {
    int lo, hi;
    s3d_sweep_shard_range(n_pairs, R, r, &lo, &hi);
    if (hi > lo) std::memcpy(records + lo, all.data() + (size_t)r * per, (size_t)(hi - lo) * rec_bytes);
  }
  return S3D_STATUS_OK;
} catch (...) { return fail_current(nullptr, sw ? &sw->err : nullptr); }

}  // extern "C"
