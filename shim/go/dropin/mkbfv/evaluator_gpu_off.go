// evaluator_gpu_off.go -- without the mkhe_gpu build tag the names of evaluator_gpu.go stand for the reference's own evaluator (see
// shim/go/dropin/mkckks/evaluator_gpu_off.go).
//
//go:build !mkhe_gpu
// +build !mkhe_gpu

package mkbfv

type GPUEvaluator = Evaluator

func NewGPUEvaluator(params Parameters) *GPUEvaluator { return NewEvaluator(params) }
