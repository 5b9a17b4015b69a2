// op_stack.cpp -- host-side mirror of the reference's operator runtime and its
// "stack" operator on top of the C ABI (include/nlstack.h).  Restates the
// bookkeeping of internal/ops/operator.go:73-166 and
// internal/ops/stack/stack.go:75-270; no pixel arithmetic happens here.
#include "op_stack.hpp"

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <thread>

#include "../../include/nlstack.h"

namespace nightlight {

// ---- internal/fits/fits.go:66-90 -------------------------------------------
ImagePtr NewImageFromNaxisn(const std::vector<int32_t> &naxisn, std::vector<float> data)
{
    int32_t pixels = 1;
    for (int32_t n : naxisn) pixels *= n;
    auto img = std::make_shared<Image>();
    img->Naxisn = naxisn;
    img->Pixels = pixels;
    if (data.empty()) data.assign((size_t)pixels, 0.0f);
    img->Data = std::move(data);
    return img;
}

// ---- internal/ops/operator.go:119-131 ----------------------------------------
std::vector<ImagePtr> RemoveNils(std::vector<ImagePtr> lights)
{
    lights.erase(std::remove(lights.begin(), lights.end(), nullptr), lights.end());
    return lights;
}

// ---- internal/ops/operator.go:73-116 ------------------------------------------
std::vector<ImagePtr> MaterializeAll(const std::vector<Promise> &ins, int maxThreads, bool forget,
                                     std::string *err)
{
    if (err) err->clear();
    if (ins.empty()) return {};
    if (maxThreads < 1) maxThreads = 1;
    std::vector<ImagePtr> outs(forget ? 0 : ins.size());
    std::vector<std::string> errs(ins.size());
    std::mutex mu;
    std::condition_variable cv;
    int running = 0;
    std::vector<std::thread> threads;
    threads.reserve(ins.size());
    for (size_t i = 0; i < ins.size(); i++) {
        {
            std::unique_lock<std::mutex> lk(mu);       // the buffered-channel limiter
            cv.wait(lk, [&] { return running < maxThreads; });
            running++;
        }
        threads.emplace_back([&, i] {
            Result r = ins[i]();
            if (!r.err.empty()) errs[i] = r.err;
            else if (!forget) outs[i] = r.image;
            std::lock_guard<std::mutex> lk(mu);
            running--;
            cv.notify_one();
        });
    }
    for (auto &t : threads) t.join();
    std::string all;                                   // distinct messages joined with "; "
    for (const auto &e : errs) {
        if (e.empty()) continue;
        if (all.empty()) all = e;
        else if (all == e) continue;
        else all = all + "; " + e;
    }
    if (err) *err = all;
    return RemoveNils(std::move(outs));
}

// ---- internal/ops/operator.go:147-166 -----------------------------------------
static std::map<std::string, OperatorFactory> &factories()
{
    static std::map<std::string, OperatorFactory> m;
    return m;
}

OperatorFactory GetOperatorFactory(const std::string &type)
{
    auto it = factories().find(type);
    return it == factories().end() ? OperatorFactory() : it->second;
}

void SetOperatorFactory(OperatorFactory f)
{
    const std::string t = f()->GetType();
    if (GetOperatorFactory(t)) throw std::logic_error("error: re-registering operator key " + t + "\n");
    factories()[t] = std::move(f);
}

// ---- internal/ops/stack/stack.go:75-99 -------------------------------------------
std::shared_ptr<OpStack> NewOpStack(int mode, int weighting, float sigmaLow, float sigmaHigh)
{
    auto op = std::make_shared<OpStack>();
    op->Type = "stack";
    op->Mode = mode;
    op->Weighting = weighting;
    op->SigmaLow = sigmaLow;
    op->SigmaHigh = sigmaHigh;
    op->RefFrameLoc = 0;
    return op;
}

std::shared_ptr<OpStack> NewOpStackDefault() { return NewOpStack(StAuto, StWeightNone, 2.75f, 2.75f); }

void RegisterOpStack()
{
    if (!GetOperatorFactory("stack"))
        SetOperatorFactory([]() -> std::shared_ptr<Operator> { return NewOpStackDefault(); });
}

static std::string fmt_g(float v)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%g", (double)v);      // Go's %g prints the shortest form too
    return buf;
}

std::string OpStack::MarshalJSON() const
{
    std::ostringstream o;
    o << "{\"type\":\"" << Type << "\",\"mode\":" << Mode << ",\"weighting\":" << Weighting
      << ",\"sigmaLow\":" << fmt_g(SigmaLow) << ",\"sigmaHigh\":" << fmt_g(SigmaHigh) << "}";
    return o.str();
}

// flat JSON object with number / string values -- all the stack operator has
static bool find_value(const std::string &s, const std::string &key, std::string *out)
{
    const std::string pat = "\"" + key + "\"";
    size_t p = s.find(pat);
    if (p == std::string::npos) return false;
    p = s.find(':', p + pat.size());
    if (p == std::string::npos) return false;
    p++;
    while (p < s.size() && isspace((unsigned char)s[p])) p++;
    size_t e = p;
    if (p < s.size() && s[p] == '"') {
        e = s.find('"', p + 1);
        if (e == std::string::npos) return false;
        *out = s.substr(p + 1, e - p - 1);
        return true;
    }
    while (e < s.size() && s[e] != ',' && s[e] != '}' && !isspace((unsigned char)s[e])) e++;
    *out = s.substr(p, e - p);
    return !out->empty();
}

bool OpStack::UnmarshalJSON(const std::string &data, std::string *err)
{
    OpStack def = *NewOpStackDefault();
    std::string v;
    size_t open = data.find('{'), close = data.rfind('}');
    if (open == std::string::npos || close == std::string::npos || close < open) {
        if (err) *err = "invalid character in JSON for operator stack";
        return false;
    }
    char *end = nullptr;
    if (find_value(data, "type", &v)) def.Type = v;
    if (find_value(data, "mode", &v)) def.Mode = (int)strtol(v.c_str(), &end, 10);
    if (find_value(data, "weighting", &v)) def.Weighting = (int)strtol(v.c_str(), &end, 10);
    if (find_value(data, "sigmaLow", &v)) def.SigmaLow = strtof(v.c_str(), &end);
    if (find_value(data, "sigmaHigh", &v)) def.SigmaHigh = strtof(v.c_str(), &end);
    Type = def.Type; Mode = def.Mode; Weighting = def.Weighting;
    SigmaLow = def.SigmaLow; SigmaHigh = def.SigmaHigh; RefFrameLoc = 0;
    return true;
}

// ---- stack.go:102-111 ----------------------------------------------------------------
std::vector<Promise> OpStack::MakePromises(const std::vector<Promise> &ins, Context *c, std::string *err)
{
    if (ins.empty()) {
        if (err) *err = Type + " operator needs inputs";
        return {};
    }
    if (err) err->clear();
    auto self = this;
    Promise out = [self, ins, c]() -> Result {
        std::string e;
        std::vector<ImagePtr> fs = MaterializeAll(ins, c->MaxThreads, false, &e);
        if (!e.empty()) return {nullptr, e};
        return self->Apply(fs, c);
    };
    return {out};
}

// ---- stack.go:231-270 ------------------------------------------------------------------
std::vector<float> getWeights(const std::vector<ImagePtr> &f, int weighting, std::string *err)
{
    if (err) err->clear();
    std::vector<float> per(f.size()), w(f.size());
    if (weighting == StWeightNone) return {};
    if (weighting == StWeightExposure) {
        for (size_t i = 0; i < f.size(); i++) per[i] = f[i]->Exposure;
    } else if (weighting == StWeightInverseNoise) {
        for (size_t i = 0; i < f.size(); i++) {
            if (!f[i]->Stats) {
                if (err) *err = std::to_string(f[i]->ID) + ": Missing stats information for noise-weighted stacking";
                return {};
            }
            per[i] = f[i]->Stats->Noise();
        }
    } else if (weighting == StWeightInverseHFR) {
        for (size_t i = 0; i < f.size(); i++) per[i] = f[i]->HFR;
    }
    int bad = -1;
    int rc = nl_weights_from_scalars(weighting, per.data(), (int)per.size(), w.data(), &bad);
    if (rc == NL_ERR_MISSING_EXPOSURE) {
        if (err) *err = std::to_string(f[(size_t)bad]->ID) + ": Missing exposure information for exposure-weighted stacking";
        return {};
    }
    if (rc != NL_OK) {
        if (err) *err = nl_last_error();
        return {};
    }
    return w;
}

static int autoSelectStackingMode(int l)   // stack.go:45-55
{
    if (l >= 25) return StLinearFit;
    if (l >= 15) return StWinsorSigma;
    if (l >= 6) return StSigma;
    return StMean;
}

// Devices the operator stacks on (not in the reference: its Apply fans out over goroutines,
// stack.go:142-152; here the same pixel-range split goes over these GPUs through nl_group_*).
// The Go shim has the same package variable.  A device may be listed more than once.
static std::vector<int> &host_devices()
{
    static std::vector<int> d;
    return d;
}

static std::vector<int> devices_for(const Context *c)
{
    if (c && !c->Devices.empty()) return c->Devices;
    if (!host_devices().empty()) return host_devices();
    return {c ? c->Device : 0};
}

// ---- stack.go:115-227 ------------------------------------------------------------------
Result OpStack::Apply(const std::vector<ImagePtr> &f, Context *c)
{
    int mode = Mode;
    if (mode < StMedian || mode > StAuto) return {nullptr, "invalid stacking mode"};
    if (f.empty()) return {nullptr, Type + " operator needs inputs"};
    if (mode == StAuto) mode = autoSelectStackingMode((int)f.size());
    if (c && c->Log) {
        char line[256];
        snprintf(line, sizeof line, "Stacking %d frames with stacking mode %d and sigma low %g high %g:\n",
                 (int)f.size(), mode, (double)SigmaLow, (double)SigmaHigh);
        *c->Log << line;
    }

    std::string err;
    std::vector<float> weights = getWeights(f, Weighting, &err);
    if (!err.empty()) return {nullptr, err};
    if (Weighting != StWeightNone && weights.empty() && Weighting > StWeightInverseHFR) {
        char msg[64];
        snprintf(msg, sizeof msg, "Invalid weighting mode %d\n", Weighting);
        return {nullptr, msg};
    }

    const std::vector<int32_t> &naxisn = f[0]->Naxisn;
    const int width = naxisn.empty() ? (int)f[0]->Data.size() : naxisn[0];
    const int height = width > 0 ? (int)(f[0]->Data.size() / (size_t)width) : 0;
    // one group per Apply, or the caller's resident one (OpStackBatches keeps buffers and the
    // stack-of-stacks accumulator on the devices across batches)
    nl_group_t *h = Resident;
    if (!h) {
        const std::vector<int> devs = devices_for(c);
        h = nl_group_create((int)f.size(), width, height, (int)devs.size(), devs.data());
    }
    if (!h) return {nullptr, nl_last_error()};
    Result out;
    do {
        int rc = Resident ? nl_group_set_active_frames(h, (int)f.size()) : NL_OK;
        for (size_t i = 0; i < f.size() && rc == NL_OK; i++) {
            if (f[i]->Data.size() != f[0]->Data.size()) { out.err = "frames differ in size"; break; }
            // a resident group was sized by an earlier batch: a frame of another size would be read
            // past its end by the per-tile row copies
            if (Resident && (int64_t)f[i]->Data.size() != ResidentPixels) { out.err = "frames differ in size"; break; }
            rc = nl_group_upload_frame(h, (int)i, f[i]->Data.data());       // overlapped; pointer not retained
        }
        if (!out.err.empty()) break;
        if (rc == NL_OK) rc = nl_group_set_weights(h, weights.empty() ? nullptr : weights.data());
        // with a resident group the result stays on the devices (Data left empty)
        std::vector<float> data(Resident ? 0 : f[0]->Data.size());
        int64_t clipLow = 0, clipHigh = 0;
        if (rc == NL_OK) rc = nl_group_run(h, mode, SigmaLow, SigmaHigh, RefFrameLoc, Resident ? nullptr : data.data(), &clipLow, &clipHigh);
        if (rc != NL_OK) { out.err = nl_last_error(); break; }
        if (mode >= StSigma && c && c->Log) {          // stack.go:214-218
            const float total = (float)((int64_t)f[0]->Data.size() * (int64_t)f.size());
            char line[256];
            snprintf(line, sizeof line, "Clipped low %lld (%.2f%%) high %lld (%.2f%%)\n",
                     (long long)clipLow, (double)((float)clipLow * 100.0f / total),
                     (long long)clipHigh, (double)((float)clipHigh * 100.0f / total));
            *c->Log << line;
        }
        float exposureSum = 0;
        for (const auto &l : f) exposureSum += l->Exposure;
        if (Resident) {
            out.image = std::make_shared<Image>();            // metadata only; pixels are on the devices
            out.image->Naxisn = naxisn;
            out.image->Pixels = (int32_t)f[0]->Data.size();
        } else {
            out.image = NewImageFromNaxisn(naxisn, std::move(data));
        }
        out.image->Exposure = exposureSum;
    } while (false);
    if (!Resident) nl_group_destroy(h);
    return out;
}

}  // namespace nightlight

// ---- OpStackBatches (internal/ops/stack/stackbatches.go) ----------------------------
namespace nightlight {

std::shared_ptr<OpStackBatches> NewOpStackBatches(std::shared_ptr<OpStack> perBatch)
{
    auto op = std::make_shared<OpStackBatches>();
    op->Type = "stackBatches";
    op->PerBatch = std::move(perBatch);
    return op;
}

std::vector<Promise> OpStackBatches::MakePromises(const std::vector<Promise> &ins, Context *c, std::string *err)
{
    if (ins.empty()) { *err = "No frames to batch process"; return {}; }          // :47-49
    std::vector<Promise> copy = ins;
    return {[this, copy, c]() { return Apply(copy, c); }};
}

// stack.go:924-937: the first light seeds the stack (copy scaled by weight), later ones add
ImagePtr StackIncremental(ImagePtr stack, const ImagePtr &light, float weight)
{
    if (!stack) {
        stack = std::make_shared<Image>(*light);                    // fits.NewImageFromImage
        for (size_t i = 0; i < light->Data.size(); i++) stack->Data[i] = light->Data[i] * weight;
    } else {
        stack->Exposure += light->Exposure;
        for (size_t i = 0; i < light->Data.size(); i++) {
            const float t = light->Data[i] * weight;                // product rounded, then added (no FMA)
            stack->Data[i] += t;
        }
    }
    return stack;
}

// stack.go:940-944 (the extended statistics are recomputed lazily by whoever needs them)
void StackIncrementalFinalize(const ImagePtr &stack, float weightSum)
{
    const float factor = 1.0f / weightSum;
    for (float &d : stack->Data) d = d * factor;
}

bool OpStackBatches::partition(const std::vector<Promise> &ins, Context *c, std::vector<Promise> *insPerm,
                               int64_t *numBatches, int64_t *batchSize, int64_t *maxThreads, std::string *err)
{
    const int64_t numFrames = (int64_t)ins.size();
    if (ins.empty()) { *err = "No input files to prepare batches"; return false; }            // :137
    Result first = ins[0]();                                                                    // :130
    if (!first.err.empty()) { *err = first.err; return false; }
    if (!first.image) { *err = "No input files to prepare batches"; return false; }
    char line[512];
    snprintf(line, sizeof line, "\nEstimating memory needs for %lld images from %s:\n", (long long)numFrames,
             first.image->FileName.c_str());
    if (c->Log) *c->Log << line;
    // the reference's estimate reads Naxisn[0] and Naxisn[1] only (stackbatches.go:130-141); its stack is generic
    // over len(Data)
    if (first.image->Naxisn.size() < 2 || first.image->Naxisn[0] <= 0 || first.image->Naxisn[1] <= 0) {
        *err = "First image has no two-dimensional shape to estimate memory needs from";
        return false;
    }
    const int64_t width = first.image->Naxisn[0], height = first.image->Naxisn[1];
    // what sizes a RESIDENT device group (more than one batch, Apply below): every sample of a frame as rows of
    // `width` -- a cube (NAXIS = 3) becomes a taller image, as the reference's flat Data is
    FirstWidth = (int)width;
    FirstHeight = (int64_t)first.image->Data.size() % width == 0 ? (int)((int64_t)first.image->Data.size() / width) : 0;
    const int64_t pixels = width * height;
    const float mPixels = (float)width * (float)height * 1e-6f;
    const int64_t bytes = pixels * 4;
    const int64_t mib = bytes / 1024 / 1024;
    snprintf(line, sizeof line,
             "%lld images of %lldx%lld pixels (%.1f MPixels), which each take %lld MiB in-memory as floating point.\n",
             (long long)numFrames, (long long)width, (long long)height, mPixels, (long long)mib);
    if (c->Log) *c->Log << line;

    const int64_t availableFrames = ((int64_t)c->StackMemoryMB * 1024 * 1024) / bytes;         // :148
    int64_t mt = c->MaxThreads > 0 ? c->MaxThreads : 1;                                         // runtime.GOMAXPROCS(0)
    snprintf(line, sizeof line,
             "CPU has %lld threads. Physical memory is %d MiB, -op.Memory is %d MiB, this fits %lld frames.\n",
             (long long)mt, c->MemoryMB, c->StackMemoryMB, (long long)availableFrames);
    if (c->Log) *c->Log << line;

    int64_t bs = 0, nb = 0;
    for (; mt >= 1; mt--) {                                                                     // :154-180
        bs = availableFrames - mt;             // lights + one temp frame per thread (no dark / flat here)
        if (bs < 2) continue;
        nb = (numFrames + bs - 1) / bs;
        if (nb > 1) bs -= 2;                   // reference frame from batch 0, and the stack of stacks
        if (bs < 2) continue;
        if (bs < mt) continue;
        break;
    }
    if (mt < 1 || bs < 2) {
        *err = "Cannot find a stacking execution path within the given memory constraints.";
        return false;
    }
    for (; (bs - 1) * nb >= numFrames; bs--) {}                                                 // :185-186
    snprintf(line, sizeof line, "Using %lld random batches of size %lld with %lld images in parallel.\n",
             (long long)nb, (long long)bs, (long long)mt);
    if (c->Log) *c->Log << line;

    *insPerm = ins;
    LastPerm.resize(ins.size());
    for (size_t i = 0; i < ins.size(); i++) LastPerm[i] = (int)i;
    if (nb > 1) {                                                                               // :190-214
        if (c->Log) *c->Log << "Randomizing input files into batches...\n";
        // rand.Perm: a uniformly random permutation of the indices (Go's global generator; here
        // a fixed-seed Fisher-Yates, the batch membership is a free choice) ...
        uint64_t state = 0x9E3779B97F4A7C15ull;
        for (size_t i = LastPerm.size(); i > 1; i--) {
            state = state * 6364136223846793005ull + 1442695040888963407ull;
            const size_t j = (size_t)((state >> 33) % i);
            std::swap(LastPerm[i - 1], LastPerm[j]);
        }
        // ... then sort.Ints inside every batch: frames keep their original relative order
        // within a batch, which fixes every frame-order fp32 sum of the per-batch stack
        for (int64_t i = 0; i < nb; i++) {
            const int64_t from = i * bs, to = std::min<int64_t>((i + 1) * bs, (int64_t)LastPerm.size());
            std::sort(LastPerm.begin() + from, LastPerm.begin() + to);
        }
        for (size_t i = 0; i < ins.size(); i++) (*insPerm)[i] = ins[(size_t)LastPerm[i]];
    }
    *numBatches = nb; *batchSize = bs; *maxThreads = mt;
    return true;
}

Result OpStackBatches::Apply(const std::vector<Promise> &ins, Context *c)
{
    std::vector<Promise> insPerm;
    int64_t numBatches = 0, batchSize = 0, maxThreads = 0;
    std::string err;
    if (!partition(ins, c, &insPerm, &numBatches, &batchSize, &maxThreads, &err)) return {nullptr, err};
    c->MaxThreads = (int)maxThreads;                                                            // :62
    c->StatsTotal = (int)insPerm.size();
    c->StatsProcessed = 0;

    // more than one batch: the frame buffers, the per-batch result and the stack of stacks stay on
    // the devices (nl_group_accumulate = StackIncremental, stack.go:924-937); only the final
    // image comes back.  The first frame sizes the group.
    nl_group_t *resident = nullptr;
    struct Guard {
        nl_group_t **g; std::shared_ptr<OpStack> per;
        ~Guard() { if (per) { per->Resident = nullptr; per->ResidentPixels = 0; } if (*g) nl_group_destroy(*g); }
    } guard{&resident, PerBatch};
    if (numBatches > 1 && PerBatch) {
        // sized by the frame partition() already loaded (the promise is not run a second time)
        const std::vector<int> devs = devices_for(c);
        if (FirstHeight <= 0) return {nullptr, "Frame data is not a whole number of image rows: cannot size the device buffers"};
        resident = nl_group_create((int)batchSize, FirstWidth, FirstHeight, (int)devs.size(), devs.data());
        if (!resident) return {nullptr, nl_last_error()};
        PerBatch->Resident = resident;
        PerBatch->ResidentPixels = (int64_t)FirstWidth * FirstHeight;
    }

    ImagePtr stack;
    int64_t stackFrames = 0;
    for (int64_t b = 0; b < numBatches; b++) {                                                  // :69
        const int64_t start = b * batchSize;
        const int64_t end = std::min<int64_t>((b + 1) * batchSize, (int64_t)insPerm.size());
        const int64_t batchFrames = end - start;
        std::vector<Promise> insBatch(insPerm.begin() + start, insPerm.begin() + end);
        char line[160];
        snprintf(line, sizeof line, "\nStarting batch %lld of %lld with %zu frames...\n", (long long)(b + 1),
                 (long long)numBatches, insBatch.size());
        if (c->Log) *c->Log << line;
        if (!PerBatch) return {nullptr, "Missing batch parameters"};                            // :82-84
        std::vector<Promise> batchPromises = PerBatch->MakePromises(insBatch, c, &err);
        if (!err.empty()) return {nullptr, err};
        if (batchPromises.size() != 1) return {nullptr, "stacking returned more than one promise"};
        Result batch = batchPromises[0]();                                                      // :92
        if (!batch.err.empty()) return {nullptr, batch.err};
        if (numBatches > 1) {                                                                   // :98-103
            // StackIncremental on the devices; the host keeps the metadata (first batch seeds
            // the stack, later ones add their exposure: stack.go:926-931)
            if (nl_group_accumulate(resident, (float)batchFrames, stack ? 0 : 1) != NL_OK)
                return {nullptr, nl_last_error()};
            if (!stack) stack = batch.image;
            else stack->Exposure += batch.image->Exposure;
            stackFrames += batchFrames;
        } else {
            stack = batch.image;
        }
    }
    if (numBatches > 1) {                                                                       // :113-116
        stack->Data.assign((size_t)stack->Pixels, 0.0f);
        if (nl_group_accumulate_finalize(resident, (float)stackFrames, stack->Data.data()) != NL_OK)
            return {nullptr, nl_last_error()};
    }
    return {stack, ""};
}

}  // namespace nightlight


// ---- plain-C entry used by the tests (and by other hosts that only have
// host buffers): decode the operator from JSON exactly as OpSequence would
// (operator.go:484-513: factory lookup by "type", then UnmarshalJSON), run it
// through MakePromises on frames supplied as promises.
extern "C" int nl_host_op_stack_apply_json(const char *json, int n_frames, int width, int height,
                                           const float *const *frames, const float *exposure,
                                           const float *hfr, int device, int max_threads,
                                           float *out, float *exposure_out,
                                           char *log_buf, int log_cap, char *err_buf, int err_cap)
{
    using namespace nightlight;
    auto put = [](char *dst, int cap, const std::string &s) {
        if (dst && cap > 0) { snprintf(dst, (size_t)cap, "%s", s.c_str()); }
    };
    put(err_buf, err_cap, "");
    put(log_buf, log_cap, "");
    RegisterOpStack();
    std::string type, err;
    const std::string js = json ? json : "{}";
    if (!find_value(js, "type", &type)) type = "stack";
    OperatorFactory fac = GetOperatorFactory(type);
    if (!fac) { put(err_buf, err_cap, "Unknown operator type '" + type + "'"); return 1; }
    std::shared_ptr<Operator> op = fac();
    auto *st = dynamic_cast<OpStack *>(op.get());
    if (!st || !st->UnmarshalJSON(js, &err)) { put(err_buf, err_cap, err); return 1; }

    std::ostringstream log;
    Context ctx;
    ctx.Log = &log;
    ctx.MaxThreads = max_threads > 0 ? max_threads : 1;
    ctx.Device = device;
    std::vector<Promise> ins;
    for (int i = 0; i < n_frames; i++) {
        ins.push_back([=]() -> Result {
            if (!frames[i]) return {nullptr, ""};          // (nil, nil): frame skipped upstream
            std::vector<float> d(frames[i], frames[i] + (size_t)width * height);
            ImagePtr img = NewImageFromNaxisn({width, height}, std::move(d));
            img->ID = i;
            img->Exposure = exposure ? exposure[i] : 0.0f;
            img->HFR = hfr ? hfr[i] : 0.0f;
            return {img, ""};
        });
    }
    std::vector<Promise> outs = op->MakePromises(ins, &ctx, &err);
    if (!err.empty()) { put(err_buf, err_cap, err); return 1; }
    if (outs.size() != 1) { put(err_buf, err_cap, "stacking returned more than one promise"); return 1; }
    Result r = outs[0]();
    put(log_buf, log_cap, log.str());
    if (!r.err.empty() || !r.image) { put(err_buf, err_cap, r.err.empty() ? "no result" : r.err); return 1; }
    if (out) memcpy(out, r.image->Data.data(), r.image->Data.size() * sizeof(float));
    if (exposure_out) *exposure_out = r.image->Exposure;
    return 0;
}

// devices every operator of this process stacks on from now on (n <= 0: back to Context.Device)
extern "C" int nl_host_set_devices(const int *devices, int n)
{
    if (n > 0 && devices) nightlight::host_devices().assign(devices, devices + n);
    else nightlight::host_devices().clear();
    return 0;
}

extern "C" const char *nl_host_op_stack_roundtrip_json(const char *json)
{
    using namespace nightlight;
    static thread_local std::string out;
    auto op = NewOpStackDefault();
    std::string err;
    if (!op->UnmarshalJSON(json ? json : "{}", &err)) { out = "error: " + err; return out.c_str(); }
    out = op->MarshalJSON();
    return out.c_str();
}


// OpStackBatches with the given per-batch operator (JSON of type "stack") over frames
// supplied as promises; stack_memory_mb steers the partition (operator.go:41).
extern "C" int nl_host_op_stack_batches_apply_json(const char *per_batch_json, int n_frames, int width, int height,
                                                   const float *const *frames, const float *exposure,
                                                   int device, int max_threads, int memory_mb, int stack_memory_mb,
                                                   float *out, float *exposure_out, int *perm_out,
                                                   char *log_buf, int log_cap, char *err_buf, int err_cap)
{
    using namespace nightlight;
    auto put = [](char *dst, int cap, const std::string &s) {
        if (dst && cap > 0) { snprintf(dst, (size_t)cap, "%s", s.c_str()); }
    };
    put(err_buf, err_cap, "");
    put(log_buf, log_cap, "");
    auto per = NewOpStackDefault();
    std::string err;
    if (!per->UnmarshalJSON(per_batch_json ? per_batch_json : "{}", &err)) { put(err_buf, err_cap, err); return 1; }
    auto op = NewOpStackBatches(per);
    std::ostringstream log;
    Context ctx;
    ctx.Log = &log;
    ctx.MaxThreads = max_threads > 0 ? max_threads : 1;
    ctx.MemoryMB = memory_mb;
    ctx.StackMemoryMB = stack_memory_mb;
    ctx.Device = device;
    std::vector<Promise> ins;
    for (int i = 0; i < n_frames; i++) {
        ins.push_back([=]() -> Result {
            std::vector<float> d(frames[i], frames[i] + (size_t)width * height);
            ImagePtr img = NewImageFromNaxisn({width, height}, std::move(d));
            img->ID = i;
            img->FileName = "frame" + std::to_string(i) + ".fits";
            img->Exposure = exposure ? exposure[i] : 0.0f;
            return {img, ""};
        });
    }
    std::vector<Promise> outs = op->MakePromises(ins, &ctx, &err);
    if (!err.empty()) { put(err_buf, err_cap, err); put(log_buf, log_cap, log.str()); return 1; }
    Result r = outs[0]();
    put(log_buf, log_cap, log.str());
    if (perm_out) for (size_t i = 0; i < op->LastPerm.size() && i < (size_t)n_frames; i++) perm_out[i] = op->LastPerm[i];
    if (!r.err.empty() || !r.image) { put(err_buf, err_cap, r.err.empty() ? "no result" : r.err); return 1; }
    if (out) memcpy(out, r.image->Data.data(), r.image->Data.size() * sizeof(float));
    if (exposure_out) *exposure_out = r.image->Exposure;
    return 0;
}
