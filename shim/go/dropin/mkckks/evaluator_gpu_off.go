// evaluator_gpu_off.go -- without the mkhe_gpu build tag the names of evaluator_gpu.go stand for the reference's own evaluator, so that tests
// patched to construct a GPUEvaluator (shim/go/patches/mkckks_tests_gpu_evaluator.diff) still build and run as the pure-Go reference.
//
//go:build !mkhe_gpu
// +build !mkhe_gpu

package mkckks

type GPUEvaluator = Evaluator

func NewGPUEvaluator(params Parameters) *GPUEvaluator { return NewEvaluator(params) }
