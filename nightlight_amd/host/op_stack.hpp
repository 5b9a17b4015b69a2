// op_stack.hpp -- C++ mirror of the reference's stack operator
// (internal/ops/stack/stack.go:33-227) on top of the C ABI (include/nlstack.h).
// Same type string, JSON fields, defaults, log lines and error messages; the
// per-pixel work runs on the MI355X through libnlstack.so.
#pragma once
#include "operator.hpp"

struct nl_group;   // include/nlstack.h (C ABI): one stack over several devices

namespace nightlight {

enum StackMode { StMedian = 0, StMean, StSigma, StWinsorSigma, StMADSigma, StLinearFit, StAuto };   // stack.go:33-42
enum StackWeighting { StWeightNone = 0, StWeightExposure, StWeightInverseNoise, StWeightInverseHFR };  // :57-63

struct OpStack : Operator, OpBase {
    int Mode = StAuto;            // json:"mode"
    int Weighting = StWeightNone; // json:"weighting"
    float SigmaLow = 2.75f;       // json:"sigmaLow"
    float SigmaHigh = 2.75f;      // json:"sigmaHigh"
    float RefFrameLoc = 0;        // json:"-"  (never assigned in the reference)
    // not in the reference: a group of device handles the caller keeps across several Apply calls
    // (OpStackBatches).  When set, Apply stacks on it and leaves the result tile on the devices
    // (the returned image carries metadata only) for nl_group_accumulate.
    ::nl_group *Resident = nullptr;
    int64_t ResidentPixels = 0;           // width*height the resident group was created for (every frame must match)

    std::string GetType() const override { return Type; }
    // stack.go:102-111: N inputs -> exactly one output promise
    std::vector<Promise> MakePromises(const std::vector<Promise> &ins, Context *c,
                                      std::string *err) override;
    // stack.go:115-227
    Result Apply(const std::vector<ImagePtr> &f, Context *c);

    std::string MarshalJSON() const;
    // stack.go:92-99: missing entries keep the defaults of NewOpStackDefault
    bool UnmarshalJSON(const std::string &data, std::string *err);
};

std::shared_ptr<OpStack> NewOpStack(int mode, int weighting, float sigmaLow, float sigmaHigh);   // stack.go:79-89
std::shared_ptr<OpStack> NewOpStackDefault();                                                    // stack.go:77
// stack.go:75 -- `init()` registers the factory for JSON decoding of type "stack"
void RegisterOpStack();

// internal/ops/stack/stackbatches.go:31-217 -- stacks the inputs in memory-sized batches with
// the given per-batch stack operator and combines the batch results into a frame-count
// weighted stack of stacks (StackIncremental / StackIncrementalFinalize, stack.go:924-944).
struct OpStackBatches : Operator, OpBase {
    std::shared_ptr<OpStack> PerBatch;    // json:"perBatch"
    std::vector<int> LastPerm;            // not in the reference: input index of every position after partition()
    int FirstWidth = 0, FirstHeight = 0;  // size of the frame partition() looked at (stackbatches.go:130-141); sizes the device group

    std::string GetType() const override { return Type; }
    std::vector<Promise> MakePromises(const std::vector<Promise> &ins, Context *c,
                                      std::string *err) override;      // stackbatches.go:46-54
    Result Apply(const std::vector<Promise> &ins, Context *c);         // :56-119
    // :121-217.  rand.Perm of the indices, then sort.Ints inside every batch (:199-209), so frames
    // keep their original relative order within a batch.  The reference draws the permutation from
    // math/rand's global generator; this mirror uses a fixed-seed Fisher-Yates (batch MEMBERSHIP is
    // a free choice of the algorithm; the order inside a batch is not, and is kept).
    bool partition(const std::vector<Promise> &ins, Context *c, std::vector<Promise> *insPerm,
                   int64_t *numBatches, int64_t *batchSize, int64_t *maxThreads, std::string *err);
};
std::shared_ptr<OpStackBatches> NewOpStackBatches(std::shared_ptr<OpStack> perBatch);   // stackbatches.go:39-44

// stack.go:924-944
ImagePtr StackIncremental(ImagePtr stack, const ImagePtr &light, float weight);
void StackIncrementalFinalize(const ImagePtr &stack, float weightSum);

// getWeights, stack.go:231-270 (returns an empty vector for StWeightNone)
std::vector<float> getWeights(const std::vector<ImagePtr> &f, int weighting, std::string *err);

}  // namespace nightlight
