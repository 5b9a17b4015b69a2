// operator.hpp -- C++ mirror of the reference's operator runtime surface that
// the stacking path sits behind (internal/ops/operator.go).  The reference is
// Go; no Go toolchain exists in the build image, so the host side above the C
// ABI is written in C++ with the same names, argument meaning and error
// behaviour, and the real cgo shim is kept under go/ (see INTEGRATION.md).
//
//   Promise            operator.go:70     func() (*fits.Image, error)
//   Context            operator.go:37-56  (only the fields the path reads)
//   MaterializeAll     operator.go:73-116
//   RemoveNils         operator.go:119-131
//   Operator           operator.go:135-138
//   OpBase             operator.go:141-145
//   SetOperatorFactory operator.go:159-166 (re-registering a type panics)
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <ostream>
#include <string>
#include <vector>

namespace nightlight {

// internal/stats/stats.go:44-175 -- only what getWeights reads (stack.go:245)
struct Stats {
    std::function<float()> noise_fn;   // lazy, like Stats.Noise()
    bool have_noise = false;
    float noise = 0;
    float Noise()
    {
        if (!have_noise && noise_fn) { noise = noise_fn(); have_noise = true; }
        return noise;
    }
};

// internal/fits/fits.go:30-54 -- the fields OpStack.Apply touches
struct Image {
    int ID = 0;
    std::string FileName;
    std::vector<int32_t> Naxisn;        // fastest varying dimension first (X, Y)
    int32_t Pixels = 0;
    std::vector<float> Data;            // row-major, len = Pixels
    float Exposure = 0;
    std::shared_ptr<Stats> Stats;
    float HFR = 0;
};
using ImagePtr = std::shared_ptr<Image>;

// fits.NewImageFromNaxisn (fits.go:66-90): data moved in, allocated if empty
ImagePtr NewImageFromNaxisn(const std::vector<int32_t> &naxisn, std::vector<float> data);

// Go's (value, error) pair; err.empty() == nil error.  (nil, nil) means
// "skip this frame" (post/postprocess.go:160-161) and is dropped by RemoveNils.
struct Result {
    ImagePtr image;
    std::string err;
};
using Promise = std::function<Result()>;

struct Context {
    std::ostream *Log = nullptr;
    int MaxThreads = 1;
    int MemoryMB = 0;                   // operator.go:40
    int StackMemoryMB = 0;              // operator.go:41 (MemoryMB*7/10)
    int StatsTotal = 0, StatsProcessed = 0;
    int Device = 0;                     // not in the reference: which GPU the HIP operator uses ...
    std::vector<int> Devices;           // ... or several: one row tile of every stack per entry (nl_group_*)
};

std::vector<ImagePtr> RemoveNils(std::vector<ImagePtr> lights);
// Materializes all promises with at most maxThreads in flight; distinct error
// strings are joined with "; " (operator.go:103-114).
std::vector<ImagePtr> MaterializeAll(const std::vector<Promise> &ins, int maxThreads, bool forget,
                                     std::string *err);

struct Operator {
    virtual ~Operator() = default;
    virtual std::string GetType() const = 0;
    virtual std::vector<Promise> MakePromises(const std::vector<Promise> &ins, Context *c,
                                              std::string *err) = 0;
};

struct OpBase {
    std::string Type;                   // json:"type"
};

using OperatorFactory = std::function<std::shared_ptr<Operator>()>;
OperatorFactory GetOperatorFactory(const std::string &type);
// throws std::logic_error("error: re-registering operator key <t>") where Go panics
void SetOperatorFactory(OperatorFactory f);

}  // namespace nightlight
