// Package stack -- drop-in replacement for the reference's
// internal/ops/stack/stack.go that runs OpStack.Apply on an AMD MI355X through
// libnlstack.so (C ABI: include/nlstack.h).
//
// NOT compiled in the build image (no Go toolchain there): this file is the
// binding a Nightlight maintainer adds.  It keeps the package name, the
// operator type string "stack", the JSON fields, the constructor names and the
// Apply signature of the reference (stack.go:66-115), so cmd/nightlight/main.go
// and internal/ops/stack/stackbatches.go compile against it unchanged.  Build
// with:  go build -tags=jsoniter,hip ./cmd/nightlight   (and drop stack.go's
// Apply behind `//go:build !hip`).
//
//go:build hip

package stack

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../nightlight_amd -lnlstack -Wl,-rpath,${SRCDIR}/../../nightlight_amd
#include <stdlib.h>
#include "nlstack.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
	"unsafe"

	"github.com/mlnoga/nightlight/internal/fits"
	"github.com/mlnoga/nightlight/internal/ops"
)

// Devices lists the GPUs this process stacks on: every stack is split into one row tile per
// entry (nl_group_*, the same pixel-range split the reference makes over goroutines,
// stack.go:142-152), the clip counters are summed on the host.  nil = all visible GPUs.
var Devices []int

// lastError reads the calling OS thread's message.  nl_last_error() is thread-local and a
// goroutine may migrate between the failing cgo call and this one, so every call+error pair
// runs under runtime.LockOSThread (see Apply).
func lastError() error { return errors.New(C.GoString(C.nl_last_error())) }

// Apply stacks a set of light frames on the GPUs.  Same contract as
// internal/ops/stack/stack.go:115-227: mode validation and auto selection,
// weights from getWeights (kept in Go, stack.go:231-270), one result image with
// the summed exposure, the "Clipped low ..." log line from the counters.
func (op *OpStack) Apply(f []*fits.Image, c *ops.Context) (result *fits.Image, err error) {
	mode := op.Mode
	if mode < StMedian || mode > StAuto {
		return nil, errors.New("invalid stacking mode")
	}
	if mode == StAuto {
		mode = autoSelectStackingMode(len(f))
	}
	fmt.Fprintf(c.Log, "Stacking %d frames with stacking mode %d and sigma low %g high %g:\n",
		len(f), mode, op.SigmaLow, op.SigmaHigh)

	weights, err := getWeights(f, op.Weighting)
	if err != nil {
		return nil, err
	}
	if mode == StMADSigma && weights != nil {
		return nil, errors.New("MADSigma stacking with weights is still unimplemented") // reference panics, stack.go:185
	}

	// the reference indexes every frame with the first frame's length (stack.go:151) and would
	// panic on a short one; the C side cannot see slice lengths, so check here
	for _, l := range f {
		if len(l.Data) != len(f[0].Data) {
			return nil, fmt.Errorf("%d: frame has %d pixels, expected %d", l.ID, len(l.Data), len(f[0].Data))
		}
	}

	runtime.LockOSThread() // nl_last_error() is per OS thread
	defer runtime.UnlockOSThread()

	width, height := int(f[0].Naxisn[0]), len(f[0].Data)/int(f[0].Naxisn[0])
	var devs *C.int
	if len(Devices) > 0 {
		cdev := make([]C.int, len(Devices))
		for i, d := range Devices {
			cdev[i] = C.int(d)
		}
		devs = &cdev[0]
	}
	// One group per Apply, as the reference allocates per call (stack.go:131-138).  nl_group_destroy parks the large
	// device buffers, the next Apply with the same geometry takes them over (bench.py "fresh_handle": create + destroy
	// 3 ms for the first handle of a process, well below one pass afterwards); C.nl_release_cached_memory() returns
	// them to the driver when the process is done stacking.
	g := C.nl_group_create(C.int(len(f)), C.int(width), C.int(height), C.int(len(Devices)), devs)
	if g == nil {
		return nil, lastError()
	}
	defer C.nl_group_destroy(g)

	// one cgo call per Go slice: [][]float32 cannot cross cgo, and the copy builds the planar
	// [N][rows*W] layout of every device tile on the way.  Each tile copies its rows into its
	// own pinned staging buffer before the call returns (no Go pointer is retained) and the
	// DMA of frame i overlaps the staging of frame i+1; the pass waits for them on the device.
	for i, l := range f {
		if rc := C.nl_group_upload_frame(g, C.int(i), (*C.float)(unsafe.Pointer(&l.Data[0]))); rc != C.NL_OK {
			return nil, lastError()
		}
	}
	var wp *C.float
	if weights != nil {
		wp = (*C.float)(unsafe.Pointer(&weights[0]))
	}
	if rc := C.nl_group_set_weights(g, wp); rc != C.NL_OK {
		return nil, lastError()
	}

	data := make([]float32, len(f[0].Data))
	var clipLow, clipHigh C.int64_t
	if rc := C.nl_group_run(g, C.int(mode), C.float(op.SigmaLow), C.float(op.SigmaHigh), C.float(op.RefFrameLoc),
		(*C.float)(unsafe.Pointer(&data[0])), &clipLow, &clipHigh); rc != C.NL_OK {
		return nil, lastError()
	}
	if mode >= StSigma {
		fmt.Fprintf(c.Log, "Clipped low %d (%.2f%%) high %d (%.2f%%)\n",
			int64(clipLow), float32(clipLow)*100.0/(float32(len(data)*len(f))),
			int64(clipHigh), float32(clipHigh)*100.0/(float32(len(data)*len(f))))
	}

	exposureSum := float32(0)
	for _, l := range f {
		exposureSum += l.Exposure
	}
	stack := fits.NewImageFromNaxisn(f[0].Naxisn, data)
	stack.Exposure = exposureSum
	return stack, nil
}
