// Package mkrlwegpu binds libmkhe_hip.so (include/mkhe.h) for the Go reference SNUCP/MKHE-KKLSS.
//
// NOT BUILT OR TESTED IN THIS REPOSITORY: the build image has no Go toolchain and lattigo v2.3.0 is not
// vendored.  It shows, entry point by entry point, the cgo binding a maintainer would add so that
// mkrlwe.KeySwitcher / mkckks.Evaluator run their hot path on an MI355X (see INTEGRATION.md).
//
//	go build -tags mkhe_gpu ./...      with CGO_CFLAGS=-I<repo>/include  CGO_LDFLAGS="-L<repo>/mkhe-kklss_amd/lib -lmkhe_hip"
//
// Round 5: this package is a LEAF -- it imports lattigo and nothing of mk-lattigo.  The reference's tests live inside its packages
// (`package mkrlwe`, `package mkckks`: they use unexported helpers), and a test of package P cannot import a package that imports P; so the
// signature-compatible drop-in types are files of the reference's own packages behind the same build tag (shim/go/dropin/mkrlwe, mkckks,
// mkbfv: they import this package), and what this package needs of mkrlwe.Parameters is the interface Params below, which that type
// satisfies as it stands.  Ciphertexts and keys cross as the lattigo values the reference's structs hold (map[string]*ring.Poly,
// []rlwe.PolyQP).
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
	"fmt"
	"runtime"
	"sort"
	"unsafe"

	"github.com/ldsec/lattigo/v2/ring"
	"github.com/ldsec/lattigo/v2/rlwe"
)

func must(rc C.int) {
	if rc != 0 {
		panic(C.GoString(C.mkhe_last_error())) // the reference panics at the same sites
	}
}

// Params is what the binding reads of mkrlwe.Parameters (mkrlwe/params.go:8-12,63-75; the rest through the embedded rlwe.Parameters).
type Params interface {
	LogN() int
	N() int
	QCount() int
	PCount() int
	MaxLevel() int
	RingQ() *ring.Ring
	RingP() *ring.Ring
	Gamma() int
	Beta(levelQ int) int
	GaloisElementForColumnRotationBy(k int) uint64
	GaloisElementForRowRotation() uint64
}

// Context replaces mkrlwe.NewKeySwitcher (mkrlwe/keyswitch.go:33-47).
type Context struct {
	c      *C.mkhe_ctx
	params Params
	ids    map[string]C.int // Go string ids -> dense ints of the C ABI
}

// psiOf returns the plain 2N-th root lattigo uses for modulus i: NttPsi[i][N/2] = psi * 2^64 mod q.
func psiOf(r *ring.Ring, i int) uint64 {
	return ring.InvMForm(r.NttPsi[i][r.N>>1], r.Modulus[i], r.MredParams[i])
}

func NewContext(params Params, device int) *Context {
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

// Params returns the parameters the context was created from.
func (ctx *Context) Params() Params { return ctx.params }

func (ctx *Context) id(s string) C.int {
	if v, ok := ctx.ids[s]; ok {
		return v
	}
	v := C.int(len(ctx.ids))
	ctx.ids[s] = v
	return v
}

// SwitchingKey mirrors mkrlwe.SwitchingKey (keys.go:23-25) on the device.  Device memory is released by Close or, failing
// that, by the finalizer; the handle keeps its Context reachable, so the context's own finalizer cannot run first.
type SwitchingKey struct {
	h   *C.mkhe_swk
	ctx *Context
}

func (k *SwitchingKey) Close() {
	if k.h != nil {
		C.mkhe_swk_destroy(k.ctx.c, k.h)
		k.h = nil
	}
	runtime.SetFinalizer(k, nil)
}

// stageSwitchingKey copies []rlwe.PolyQP (mkrlwe.SwitchingKey.Value) into ONE contiguous Go buffer [digit][Q limbs..., P limbs...][N], the layout
// of mkhe_swk_upload / mkhe_swk_download.  (Storing the Go limb pointers in a C array -- mkhe_swk_upload_limbs -- is not allowed by the cgo
// pointer rules unless every slice is pinned with runtime.Pinner; one staging copy of 56 MiB at key-upload time is the simpler contract.)
func (ctx *Context) swkWords(digits int) int {
	return digits * (ctx.params.QCount() + ctx.params.PCount()) * ctx.params.N()
}

// UploadSwitchingKey copies a switching key / CRS / hoisted digit vector (value = mkrlwe.SwitchingKey.Value, NTT domain as the Go side
// stores it) to the device.  Digits beyond len(value) -- a key allocated at a lower level -- stay zero.
func (ctx *Context) UploadSwitchingKey(value []rlwe.PolyQP) *SwitchingKey {
	out := &SwitchingKey{ctx: ctx}
	must(C.mkhe_swk_create(ctx.c, &out.h))
	runtime.SetFinalizer(out, func(k *SwitchingKey) { k.Close() })
	nq, np, n := ctx.params.QCount(), ctx.params.PCount(), ctx.params.N()
	beta := ctx.params.Beta(ctx.params.MaxLevel())
	if len(value) > beta {
		panic("mkrlwegpu: switching key with more digits than Beta(MaxLevel)")
	}
	buf := make([]uint64, ctx.swkWords(beta))
	k := 0
	for _, p := range value {
		for j := 0; j < nq; j++ {
			if j < len(p.Q.Coeffs) {
				copy(buf[k:k+n], p.Q.Coeffs[j])
			}
			k += n
		}
		for j := 0; j < np; j++ {
			if j < len(p.P.Coeffs) {
				copy(buf[k:k+n], p.P.Coeffs[j])
			}
			k += n
		}
	}
	must(C.mkhe_swk_upload(ctx.c, out.h, (*C.uint64_t)(unsafe.Pointer(&buf[0]))))
	runtime.KeepAlive(buf)
	return out
}

// DownloadSwitchingKey writes the device key back into value (the digits and limbs value has).
func (ctx *Context) DownloadSwitchingKey(d *SwitchingKey, value []rlwe.PolyQP) {
	nq, np, n := ctx.params.QCount(), ctx.params.PCount(), ctx.params.N()
	beta := ctx.params.Beta(ctx.params.MaxLevel())
	buf := make([]uint64, ctx.swkWords(beta))
	must(C.mkhe_swk_download(ctx.c, d.h, (*C.uint64_t)(unsafe.Pointer(&buf[0]))))
	k := 0
	for i := 0; i < beta; i++ {
		for j := 0; j < nq+np; j++ {
			if i < len(value) {
				if j < nq && j < len(value[i].Q.Coeffs) {
					copy(value[i].Q.Coeffs[j], buf[k:k+n])
				} else if j >= nq && j-nq < len(value[i].P.Coeffs) {
					copy(value[i].P.Coeffs[j-nq], buf[k:k+n])
				}
			}
			k += n
		}
	}
	runtime.KeepAlive(d)
}

// Ciphertext mirrors mkrlwe.Ciphertext (elements.go:17-19) on the device.
type Ciphertext struct {
	h   *C.mkhe_ct
	ids []string
	ctx *Context
}

// IDs are the party ids of the device ciphertext, in slot order (slot 0 = "0", slot 1 + i = IDs()[i]).
func (c *Ciphertext) IDs() []string { return c.ids }

func (c *Ciphertext) Close() {
	if c.h != nil {
		C.mkhe_ct_destroy(c.ctx.c, c.h)
		c.h = nil
	}
	runtime.SetFinalizer(c, nil)
}

func (ctx *Context) newCt(ids []string, level int) *Ciphertext {
	cids := make([]C.int, len(ids)+1)
	for i, s := range ids {
		cids[i] = ctx.id(s)
	}
	out := &Ciphertext{ids: ids, ctx: ctx}
	must(C.mkhe_ct_create(ctx.c, C.int(len(ids)), &cids[0], C.int(level+1), &out.h))
	runtime.SetFinalizer(out, func(c *Ciphertext) { c.Close() })
	return out
}

// SortedIDs: the party ids of a ciphertext value (every key but "0") in the fixed order the device slots use.  Any fixed order gives the
// reference's result: every accumulation of the algorithm is canonical (SURVEY.md 8b; Go iterates its maps in random order).
func SortedIDs(value map[string]*ring.Poly) []string {
	ids := make([]string, 0, len(value))
	for id := range value {
		if id != "0" {
			ids = append(ids, id)
		}
	}
	sort.Strings(ids)
	return ids
}

// Upload copies value (mkrlwe.Ciphertext.Value: "0" and one polynomial per party id, coefficient domain) at the given level to the device
// through one contiguous staging buffer [slot][limb][N].
func (ctx *Context) Upload(value map[string]*ring.Poly, level int) *Ciphertext {
	ids := SortedIDs(value)
	out := ctx.newCt(ids, level)
	limbs, n := level+1, ctx.params.N()
	buf := make([]uint64, (len(ids)+1)*limbs*n)
	for slot, id := range append([]string{"0"}, ids...) {
		for j := 0; j < limbs; j++ {
			copy(buf[(slot*limbs+j)*n:(slot*limbs+j+1)*n], value[id].Coeffs[j])
		}
	}
	must(C.mkhe_ct_upload(ctx.c, out.h, (*C.uint64_t)(unsafe.Pointer(&buf[0]))))
	runtime.KeepAlive(buf)
	return out
}

// UploadPoly: one polynomial as a ciphertext without party components (slot 0), the operand form of Decompose / ExternalProduct.
func (ctx *Context) UploadPoly(p *ring.Poly, level int) *Ciphertext {
	return ctx.Upload(map[string]*ring.Poly{"0": p}, level)
}

// Download writes the device ciphertext back into value (which must hold the same ids; level + 1 limbs of every polynomial are written).
func (ctx *Context) Download(d *Ciphertext, value map[string]*ring.Poly, level int) {
	limbs, n := level+1, ctx.params.N()
	buf := make([]uint64, (len(d.ids)+1)*limbs*n)
	must(C.mkhe_ct_download(ctx.c, d.h, (*C.uint64_t)(unsafe.Pointer(&buf[0]))))
	for slot, id := range append([]string{"0"}, d.ids...) {
		p, ok := value[id]
		if !ok {
			panic("mkrlwegpu: Download into a ciphertext that lacks id " + id)
		}
		for j := 0; j < limbs; j++ {
			copy(p.Coeffs[j], buf[(slot*limbs+j)*n:(slot*limbs+j+1)*n])
		}
	}
	runtime.KeepAlive(d)
}

// RelinKeys holds the device copies of rlkSet.Value[id].Value[0..2] = (b, d, v) (keys.go:34-37).
type RelinKeys map[string][3]*SwitchingKey

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

// MulRelinRescale is mkckks.Evaluator.mulRelinHoisted's MulAndRelin + single Rescale (mkckks/evaluator.go:558-581) as one engine call:
// out is the rescaled ciphertext, one level below the product.  hoisted0 / hoisted1 as for MulAndRelinHoisted (nil: the engine hoists).
func (ctx *Context) MulRelinRescale(op0, op1 *Ciphertext, hoisted0, hoisted1 []*SwitchingKey, rk RelinKeys, crsU *SwitchingKey, out *Ciphertext) {
	b1 := handles(op1.ids, rk, 0)
	d0 := handles(op0.ids, rk, 1)
	v0 := handles(op0.ids, rk, 2)
	var h0, h1 **C.mkhe_swk
	if hoisted0 != nil {
		h0 = swkArray(hoisted0)
		defer C.free(unsafe.Pointer(h0))
	}
	if hoisted1 != nil {
		h1 = swkArray(hoisted1)
		defer C.free(unsafe.Pointer(h1))
	}
	must(C.mkhe_mul_relin_rescale(ctx.c, op0.h, op1.h, h0, h1,
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
