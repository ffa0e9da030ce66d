// Package mkrlwegpu binds libmkhe_hip.so (include/mkhe.h) for the Go reference SNUCP/MKHE-KKLSS.
//
// NOT BUILT OR TESTED IN THIS REPOSITORY: the build image has no Go toolchain and lattigo v2.3.0 is not
// vendored.  It shows, entry point by entry point, the cgo binding a maintainer would add so that
// mkrlwe.KeySwitcher / mkckks.Evaluator run their hot path on an MI355X (see INTEGRATION.md).
//
//	go build -tags mkhe_gpu ./...      with CGO_CFLAGS=-I<repo>/include  CGO_LDFLAGS="-L<repo>/mkhe-kklss_amd/lib -lmkhe_hip"
//
//go:build mkhe_gpu

package mkrlwegpu

/*
#include <stdlib.h>
#include "mkhe.h"
*/
import "C"

import (
	"fmt"
	"runtime"
	"unsafe"

	"github.com/ldsec/lattigo/v2/ring"
	"github.com/ldsec/lattigo/v2/rlwe"

	"mk-lattigo/mkrlwe"
)

func must(rc C.int) {
	if rc != 0 {
		panic(C.GoString(C.mkhe_last_error())) // the reference panics at the same sites
	}
}

// Context replaces mkrlwe.NewKeySwitcher (mkrlwe/keyswitch.go:33-47).
type Context struct {
	c      *C.mkhe_ctx
	params mkrlwe.Parameters
	ids    map[string]C.int // Go string ids -> dense ints of the C ABI
}

// psiOf returns the plain 2N-th root lattigo uses for modulus i: NttPsi[i][N/2] = psi * 2^64 mod q.
func psiOf(r *ring.Ring, i int) uint64 {
	return ring.InvMForm(r.NttPsi[i][r.N>>1], r.Modulus[i], r.MredParams[i])
}

func NewContext(params mkrlwe.Parameters, device int) *Context {
	rq, rp := params.RingQ(), params.RingP()
	psiQ := make([]uint64, len(rq.Modulus))
	psiP := make([]uint64, len(rp.Modulus))
	for i := range psiQ {
		psiQ[i] = psiOf(rq, i)
	}
	for i := range psiP {
		psiP[i] = psiOf(rp, i)
	}
	ctx := &Context{params: params, ids: map[string]C.int{}}
	must(C.mkhe_ctx_create(&ctx.c, C.int(params.LogN()),
		(*C.uint64_t)(unsafe.Pointer(&rq.Modulus[0])), C.int(len(rq.Modulus)),
		(*C.uint64_t)(unsafe.Pointer(&rp.Modulus[0])), C.int(len(rp.Modulus)),
		C.int(params.Gamma()),
		(*C.uint64_t)(unsafe.Pointer(&psiQ[0])), (*C.uint64_t)(unsafe.Pointer(&psiP[0])), C.int(device)))
	runtime.SetFinalizer(ctx, func(c *Context) { C.mkhe_ctx_destroy(c.c) })
	return ctx
}

func (ctx *Context) id(s string) C.int {
	if v, ok := ctx.ids[s]; ok {
		return v
	}
	v := C.int(len(ctx.ids))
	ctx.ids[s] = v
	return v
}

// SwitchingKey mirrors mkrlwe.SwitchingKey (keys.go:23-25) on the device.
type SwitchingKey struct{ h *C.mkhe_swk }

// UploadSwitchingKey copies []rlwe.PolyQP limb by limb (cgo cannot pass [][]uint64 directly: the limb
// pointers are collected in C memory first).
func (ctx *Context) UploadSwitchingKey(swk *mkrlwe.SwitchingKey) *SwitchingKey {
	out := &SwitchingKey{}
	must(C.mkhe_swk_create(ctx.c, &out.h))
	nq, np := ctx.params.QCount(), ctx.params.PCount()
	n := len(swk.Value) * (nq + np)
	ptrs := (*[1 << 20]*C.uint64_t)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	defer C.free(unsafe.Pointer(ptrs))
	k := 0
	for _, p := range swk.Value {
		for j := 0; j < nq; j++ {
			ptrs[k] = (*C.uint64_t)(unsafe.Pointer(&p.Q.Coeffs[j][0]))
			k++
		}
		for j := 0; j < np; j++ {
			ptrs[k] = (*C.uint64_t)(unsafe.Pointer(&p.P.Coeffs[j][0]))
			k++
		}
	}
	must(C.mkhe_swk_upload_limbs(ctx.c, out.h, (**C.uint64_t)(unsafe.Pointer(ptrs)), C.int(len(swk.Value))))
	runtime.KeepAlive(swk)
	return out
}

// Ciphertext mirrors mkrlwe.Ciphertext (elements.go:17-19) on the device.
type Ciphertext struct {
	h   *C.mkhe_ct
	ids []string
}

func (ctx *Context) newCt(ids []string, level int) *Ciphertext {
	cids := make([]C.int, len(ids)+1)
	for i, s := range ids {
		cids[i] = ctx.id(s)
	}
	out := &Ciphertext{ids: ids}
	must(C.mkhe_ct_create(ctx.c, C.int(len(ids)), &cids[0], C.int(level+1), &out.h))
	return out
}

func limbPtrs(p *ring.Poly, limbs int) unsafe.Pointer {
	ptrs := (*[1 << 16]*C.uint64_t)(C.malloc(C.size_t(limbs) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	for j := 0; j < limbs; j++ {
		ptrs[j] = (*C.uint64_t)(unsafe.Pointer(&p.Coeffs[j][0]))
	}
	return unsafe.Pointer(ptrs)
}

func sortedIDs(ct *mkrlwe.Ciphertext) []string {
	ids := []string{}
	for id := range ct.IDSet().Value {
		ids = append(ids, id)
	}
	// any fixed order: every accumulation of the algorithm is canonical (SURVEY.md 8b)
	for i := range ids {
		for j := i + 1; j < len(ids); j++ {
			if ids[j] < ids[i] {
				ids[i], ids[j] = ids[j], ids[i]
			}
		}
	}
	return ids
}

// Upload copies ct.Value[...] (coefficient domain) to the device.
func (ctx *Context) Upload(ct *mkrlwe.Ciphertext) *Ciphertext {
	ids := sortedIDs(ct)
	out := ctx.newCt(ids, ct.Level())
	for slot, id := range append([]string{"0"}, ids...) {
		p := limbPtrs(ct.Value[id], ct.Level()+1)
		must(C.mkhe_ct_upload_poly_limbs(ctx.c, out.h, C.int(slot), (**C.uint64_t)(p)))
		C.free(p)
	}
	runtime.KeepAlive(ct)
	return out
}

// Download writes the device ciphertext back into ct (which must have the same ids and level).
func (ctx *Context) Download(d *Ciphertext, ct *mkrlwe.Ciphertext) {
	for slot, id := range append([]string{"0"}, d.ids...) {
		p := limbPtrs(ct.Value[id], ct.Level()+1)
		must(C.mkhe_ct_download_poly_limbs(ctx.c, d.h, C.int(slot), (**C.uint64_t)(p)))
		C.free(p)
	}
}

// RelinKeys holds the device copies of rlkSet.Value[id].Value[0..2] = (b, d, v) (keys.go:34-37).
type RelinKeys map[string][3]*SwitchingKey

func (ctx *Context) UploadRelinKeys(rlkSet *mkrlwe.RelinearizationKeySet) RelinKeys {
	out := RelinKeys{}
	for id, rlk := range rlkSet.Value {
		out[id] = [3]*SwitchingKey{ctx.UploadSwitchingKey(rlk.Value[0]), ctx.UploadSwitchingKey(rlk.Value[1]), ctx.UploadSwitchingKey(rlk.Value[2])}
	}
	return out
}

func handles(ids []string, rk RelinKeys, which int) []*C.mkhe_swk {
	out := make([]*C.mkhe_swk, len(ids)+1)
	for i, id := range ids {
		k, ok := rk[id]
		if !ok {
			panic("cannot GetRelinearizationKey: there is no relinearization key with given id") // keys.go:190-198
		}
		out[i] = k[which].h
	}
	return out
}

// MulAndRelin replaces KeySwitcher.MulAndRelin / MulAndRelinHoisted with nil hoisted forms
// (keyswitch.go:122-230, keyswitch_hoisted.go:44-179).  crsU = params.CRS[-1] uploaded once.
func (ctx *Context) MulAndRelin(op0, op1 *Ciphertext, rk RelinKeys, crsU *SwitchingKey, out *Ciphertext) {
	b1 := handles(op1.ids, rk, 0)
	d0 := handles(op0.ids, rk, 1)
	v0 := handles(op0.ids, rk, 2)
	must(C.mkhe_mul_and_relin(ctx.c, op0.h, op1.h, nil, nil,
		(**C.mkhe_swk)(unsafe.Pointer(&b1[0])), (**C.mkhe_swk)(unsafe.Pointer(&d0[0])),
		(**C.mkhe_swk)(unsafe.Pointer(&v0[0])), crsU.h, out.h))
}

// Decompose replaces KeySwitcher.Decompose (keyswitch.go:49-73) for ct.Value[id].
func (ctx *Context) Decompose(level int, ct *Ciphertext, slot int, isNTT bool, out *SwitchingKey) {
	ntt := C.int(0)
	if isNTT {
		ntt = 1
	}
	must(C.mkhe_decompose(ctx.c, C.int(level), ntt, ct.h, C.int(slot), out.h))
}

// Rescale replaces the body of mkckks.Evaluator.Rescale (evaluator.go:385-391); the float64 loop that
// decides nbRescales (:376-384) stays in Go.
func (ctx *Context) Rescale(in *Ciphertext, nb int, out *Ciphertext) {
	must(C.mkhe_rescale(ctx.c, in.h, C.int(nb), out.h))
}

// RotateHoisted replaces KeySwitcher.RotateHoisted / Rotate (hoisted == nil) (keyswitch_hoisted.go:183-247).
func (ctx *Context) RotateHoisted(in *Ciphertext, rotidx int, hoisted []*SwitchingKey, rk []*SwitchingKey, crs *SwitchingKey, out *Ciphertext) {
	galEl := ctx.params.GaloisElementForColumnRotationBy(rotidx)
	rkh := make([]*C.mkhe_swk, len(rk)+1)
	for i, k := range rk {
		rkh[i] = k.h
	}
	var hh **C.mkhe_swk
	if hoisted != nil {
		hs := make([]*C.mkhe_swk, len(hoisted)+1)
		for i, k := range hoisted {
			hs[i] = k.h
		}
		hh = (**C.mkhe_swk)(unsafe.Pointer(&hs[0]))
	}
	must(C.mkhe_rotate(ctx.c, C.uint64_t(galEl), in.h, hh, (**C.mkhe_swk)(unsafe.Pointer(&rkh[0])), crs.h, out.h))
}

var _ = fmt.Sprintf
var _ rlwe.PolyQP
