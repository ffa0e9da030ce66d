// Package mkrlwegpu, batchgpu.go: B inputs of ONE shape per call (mkhe_*_batch; no reference counterpart -- the Go evaluator issues one operation at
// a time).  A service that evaluates one circuit on many inputs calls these instead of B single operations: every output equals the single-operation
// wrapper's on the same input bit for bit (tests/test_gpu_batch.py), and on the small rings the throughput is 3x (DESIGN.md section 8 "B inputs in
// lock step").  All inputs of a call have the same ids and level; keys and CRS belong to the parties and are shared; an operand that is the same for
// every input (a model) is passed B times.  Un-built here like the rest of the shim (no Go toolchain in the build image); its C calls are checked
// against include/mkhe.h by tests/test_go_shim_static.py.
//
//go:build mkhe_gpu
// +build mkhe_gpu

package mkrlwegpu

/*
#include <stdlib.h>
#include "mkhe.h"
*/
import "C"

import (
	"unsafe"
)

// sameLen panics unless every list of a batched call has the batch's length: C reads nbatch entries of each array (ADVICE r4).
func sameLen(what string, n int, lens ...int) {
	for _, l := range lens {
		if l != n {
			panic("mkrlwegpu: " + what + ": the lists of a batch must have one length")
		}
	}
}

// ctArray copies ciphertext handles into C memory (see swkArray).  The caller frees it.
func ctArray(cts []*Ciphertext) **C.mkhe_ct {
	arr := (**C.mkhe_ct)(C.malloc(C.size_t(len(cts)+1) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	s := ctSlice(arr, len(cts)+1)
	for i, c := range cts {
		s[i] = c.h
	}
	s[len(cts)] = nil
	return arr
}

// NewCiphertextBatch creates B ciphertexts of one shape as views of one pooled block (mkhe_ct_create_batch); Close them one by one.
func (ctx *Context) NewCiphertextBatch(ids []string, level int, B int) []*Ciphertext {
	cids := make([]C.int, len(ids)+1)
	for i, id := range ids {
		cids[i] = ctx.id(id)
	}
	arr := (**C.mkhe_ct)(C.malloc(C.size_t(B+1) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	defer C.free(unsafe.Pointer(arr))
	must(C.mkhe_ct_create_batch(ctx.c, C.int(B), C.int(len(ids)), &cids[0], C.int(level+1), arr))
	s := ctSlice(arr, B)
	out := make([]*Ciphertext, B)
	for i := range out {
		out[i] = &Ciphertext{h: s[i], ids: append([]string(nil), ids...), ctx: ctx}
	}
	return out
}

// NewSwitchingKeyBatch creates count switching keys (hoisted forms) as views of one pooled block (mkhe_swk_create_batch).
func (ctx *Context) NewSwitchingKeyBatch(count int) []*SwitchingKey {
	arr := (**C.mkhe_swk)(C.malloc(C.size_t(count+1) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	defer C.free(unsafe.Pointer(arr))
	must(C.mkhe_swk_create_batch(ctx.c, C.int(count), arr))
	s := swkSlice(arr, count)
	out := make([]*SwitchingKey, count)
	for i := range out {
		out[i] = &SwitchingKey{h: s[i], ctx: ctx}
	}
	return out
}

// HoistedFormBatch: mkckks.Evaluator.HoistedForm (mkckks/evaluator.go:543-553) of B ciphertexts; the result holds len(ids) keys per input, input
// after input, aligned with the ids.
func (ctx *Context) HoistedFormBatch(cts []*Ciphertext, level int) []*SwitchingKey {
	if len(cts) == 0 || len(cts[0].ids) == 0 {
		return nil
	}
	out := ctx.NewSwitchingKeyBatch(len(cts) * len(cts[0].ids))
	in := ctArray(cts)
	defer C.free(unsafe.Pointer(in))
	arr := swkArray(out)
	defer C.free(unsafe.Pointer(arr))
	must(C.mkhe_hoisted_form_batch(ctx.c, C.int(level), C.int(len(cts)), in, arr))
	return out
}

// MulRelinBatch: KeySwitcher.MulAndRelin[Hoisted] (mkrlwe/keyswitch_hoisted.go:44-179) for B pairs, followed by the single Rescale of
// mkckks.Evaluator.mulRelinHoisted (mkckks/evaluator.go:558-581) when rescale is set (out is then one level below the product).  hoisted0 /
// hoisted1: HoistedFormBatch results or nil (the engine hoists).
func (ctx *Context) MulRelinBatch(op0, op1 []*Ciphertext, hoisted0, hoisted1 []*SwitchingKey, rk RelinKeys, crsU *SwitchingKey, rescale bool, out []*Ciphertext) {
	if len(op0) == 0 {
		return
	}
	sameLen("MulRelinBatch", len(op0), len(op1), len(out))
	if hoisted0 != nil {
		sameLen("MulRelinBatch (hoisted0)", len(op0)*len(op0[0].ids), len(hoisted0))
	}
	if hoisted1 != nil {
		sameLen("MulRelinBatch (hoisted1)", len(op1)*len(op1[0].ids), len(hoisted1))
	}
	b1 := handles(op1[0].ids, rk, 0)
	d0 := handles(op0[0].ids, rk, 1)
	v0 := handles(op0[0].ids, rk, 2)
	a0, a1, ao := ctArray(op0), ctArray(op1), ctArray(out)
	defer C.free(unsafe.Pointer(a0))
	defer C.free(unsafe.Pointer(a1))
	defer C.free(unsafe.Pointer(ao))
	var h0, h1 **C.mkhe_swk
	if hoisted0 != nil {
		h0 = swkArray(hoisted0)
		defer C.free(unsafe.Pointer(h0))
	}
	if hoisted1 != nil {
		h1 = swkArray(hoisted1)
		defer C.free(unsafe.Pointer(h1))
	}
	must(C.mkhe_mul_relin_batch(ctx.c, C.int(len(op0)), a0, a1, h0, h1,
		(**C.mkhe_swk)(unsafe.Pointer(&b1[0])), (**C.mkhe_swk)(unsafe.Pointer(&d0[0])),
		(**C.mkhe_swk)(unsafe.Pointer(&v0[0])), crsU.h, b2i(rescale), ao))
}

// RotateBatch: KeySwitcher.RotateHoisted / Rotate (mkrlwe/keyswitch_hoisted.go:183-247, keyswitch.go:234-298) of B ciphertexts by one rotation
// index; hoisted: HoistedFormBatch result or nil; rk aligned with the ids, crs = params.CRS[rotidx].
func (ctx *Context) RotateBatch(in []*Ciphertext, rotidx int, hoisted []*SwitchingKey, rk []*SwitchingKey, crs *SwitchingKey, out []*Ciphertext) {
	if len(in) == 0 {
		return
	}
	sameLen("RotateBatch", len(in), len(out))
	sameLen("RotateBatch (rk)", len(in[0].ids), len(rk))
	if hoisted != nil {
		sameLen("RotateBatch (hoisted)", len(in)*len(in[0].ids), len(hoisted))
	}
	ai, ao := ctArray(in), ctArray(out)
	defer C.free(unsafe.Pointer(ai))
	defer C.free(unsafe.Pointer(ao))
	var h **C.mkhe_swk
	if hoisted != nil {
		h = swkArray(hoisted)
		defer C.free(unsafe.Pointer(h))
	}
	r := swkArray(rk)
	defer C.free(unsafe.Pointer(r))
	must(C.mkhe_rotate_batch(ctx.c, C.uint64_t(ctx.params.GaloisElementForColumnRotationBy(rotidx)), C.int(len(in)), ai, h, r, crs.h, ao))
}

// AddBatch / SubBatch: mkckks.Evaluator.AddNew / SubNew (mkckks/evaluator.go:316-356) for B pairs (scales already matched by the caller).
func (ctx *Context) AddBatch(op0, op1, out []*Ciphertext) { ctx.binaryBatch(0, op0, op1, out) }
func (ctx *Context) SubBatch(op0, op1, out []*Ciphertext) { ctx.binaryBatch(1, op0, op1, out) }

func (ctx *Context) binaryBatch(op int, op0, op1, out []*Ciphertext) {
	if len(op0) == 0 {
		return
	}
	sameLen("AddBatch / SubBatch", len(op0), len(op1), len(out))
	a0, a1, ao := ctArray(op0), ctArray(op1), ctArray(out)
	defer C.free(unsafe.Pointer(a0))
	defer C.free(unsafe.Pointer(a1))
	defer C.free(unsafe.Pointer(ao))
	must(C.mkhe_ct_binary_batch(ctx.c, C.int(op), C.int(len(op0)), a0, a1, ao))
}

// MulPtxtBatch: mkckks.Evaluator.MulPtxtNew (mkckks/evaluator.go:465-481) for B ciphertexts and one resident plaintext, followed by nbRescale
// DivRoundByLastModulus steps (the host decides nbRescale from the scales, :376-384).
func (ctx *Context) MulPtxtBatch(in []*Ciphertext, pt *Plaintext, nbRescale int, out []*Ciphertext) {
	if len(in) == 0 {
		return
	}
	sameLen("MulPtxtBatch", len(in), len(out))
	ai, ao := ctArray(in), ctArray(out)
	defer C.free(unsafe.Pointer(ai))
	defer C.free(unsafe.Pointer(ao))
	must(C.mkhe_ct_mul_ptxt_batch(ctx.c, C.int(len(in)), ai, pt.dev, C.int(nbRescale), ao))
}

// RotateMulti: B rotations of B ciphertexts of one shape, each by its own rotation index with its own keys (mkhe_rotate_multi; rk flat
// [b * n + a], crs[b] = params.CRS[rotidx[b]]), optionally out[b] = postAdd[b] + Rotate(in[b]) with the AddNew on the rotation's store
// (cnn/cnn.go:16-37,51-67: the independent chains of a layer as lanes of one launch set, the log-sum steps as one pass).
func (ctx *Context) RotateMulti(in []*Ciphertext, rotidx []int, hoisted []*SwitchingKey, rk []*SwitchingKey, crs []*SwitchingKey, postAdd []*Ciphertext, out []*Ciphertext) {
	if len(in) == 0 {
		return
	}
	sameLen("RotateMulti", len(in), len(out), len(rotidx), len(crs))
	sameLen("RotateMulti (rk)", len(in)*len(in[0].ids), len(rk))
	gal := make([]C.uint64_t, len(rotidx))
	for i, r := range rotidx {
		gal[i] = C.uint64_t(ctx.params.GaloisElementForColumnRotationBy(r))
	}
	ai, ao := ctArray(in), ctArray(out)
	defer C.free(unsafe.Pointer(ai))
	defer C.free(unsafe.Pointer(ao))
	var h **C.mkhe_swk
	if hoisted != nil {
		sameLen("RotateMulti (hoisted)", len(in)*len(in[0].ids), len(hoisted))
		h = swkArray(hoisted)
		defer C.free(unsafe.Pointer(h))
	}
	r, c := swkArray(rk), swkArray(crs)
	defer C.free(unsafe.Pointer(r))
	defer C.free(unsafe.Pointer(c))
	var p **C.mkhe_ct
	if postAdd != nil {
		sameLen("RotateMulti (postAdd)", len(in), len(postAdd))
		p = ctArray(postAdd)
		defer C.free(unsafe.Pointer(p))
	}
	must(C.mkhe_rotate_multi(ctx.c, C.int(len(in)), &gal[0], ai, h, r, c, p, ao))
}

// Sum: out = in[0] + in[1] + ... (mkhe_ct_sum: the AddNew chain over a layer's products, cnn/cnn.go:19-30,58-62, as one launch); out may be in[0].
func (ctx *Context) Sum(in []*Ciphertext, out *Ciphertext) {
	if len(in) == 0 {
		return
	}
	a := ctArray(in)
	defer C.free(unsafe.Pointer(a))
	must(C.mkhe_ct_sum(ctx.c, C.int(len(in)), a, out.h))
}

// SetNTTChoice pins the forward kernel of N = 2^15 for one launch shape (limbs per launch; decompose: the Decompose-fused form): choice 0 = two passes,
// 1 = single pass, -1 = measure again (mkhe_ctx_set_ntt_choice).  limbs = 0 pins every shape not measured yet.  A bench line names its choices in
// config.ntt_kernel_choice; with them pinned it can be repeated on the same kernels.
func (ctx *Context) SetNTTChoice(limbs int, decompose bool, choice int) {
	must(C.mkhe_ctx_set_ntt_choice(ctx.c, C.long(limbs), b2i(decompose), C.int(choice)))
}

// SetBatchLanes: from how many hoisted limb-NTTs per input MulRelinBatch evaluates its inputs in flight (this context and two internal ones, round
// robin) instead of in lock step, N = 2^15 (mkhe_ctx_set_batch_lanes; default 1536, 0 = always, negative = never).  Same results either way.
func (ctx *Context) SetBatchLanes(minLimbs int) {
	must(C.mkhe_ctx_set_batch_lanes(ctx.c, C.long(minLimbs)))
}

// PoolHeldBytes / PoolTrim: device memory the context's buffer pool holds for reuse, and its release (MKHE_POOL_GB bounds it per device).
func (ctx *Context) PoolHeldBytes() int64 { return int64(C.mkhe_pool_held_bytes(ctx.c)) }
func (ctx *Context) PoolTrim()            { must(C.mkhe_pool_trim(ctx.c)) }

