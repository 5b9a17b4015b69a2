// op_stack.hpp -- C++ mirror of the reference's stack operator
// (internal/ops/stack/stack.go:33-227) on top of the C ABI (include/nlstack.h).
// Same type string, JSON fields, defaults, log lines and error messages; the
// per-pixel work runs on the MI355X through libnlstack.so.
#pragma once
#include "operator.hpp"

namespace nightlight {

enum StackMode { StMedian = 0, StMean, StSigma, StWinsorSigma, StMADSigma, StLinearFit, StAuto };   // stack.go:33-42
enum StackWeighting { StWeightNone = 0, StWeightExposure, StWeightInverseNoise, StWeightInverseHFR };  // :57-63

struct OpStack : Operator, OpBase {
    int Mode = StAuto;            // json:"mode"
    int Weighting = StWeightNone; // json:"weighting"
    float SigmaLow = 2.75f;       // json:"sigmaLow"
    float SigmaHigh = 2.75f;      // json:"sigmaHigh"
    float RefFrameLoc = 0;        // json:"-"  (never assigned in the reference)

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

// getWeights, stack.go:231-270 (returns an empty vector for StWeightNone)
std::vector<float> getWeights(const std::vector<ImagePtr> &f, int weighting, std::string *err);

}  // namespace nightlight
