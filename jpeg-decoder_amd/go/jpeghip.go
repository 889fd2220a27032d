// jpeghip.go -- cgo shim that puts libjpeghip.so behind maxasm/jpeg-decoder's Go host.
//
// Drop this file next to main.go (package main).  It replaces the two call sites of decodeStartOfScan
// that SURVEY.md section 8b names:
//
//	cut 1  main.go:1022-1032  generateCodes + decodeHuffmanData(header, &BitReader{...})
//	cut 2  main.go:1054-1057  dequantize; inverseDCT; spreadCoeffecients; convertColorSpace
//
// with ONE call, decodeOnGPU(header, rawECS), that returns the cropped RGB8 image writeBitMap needs.
// The Go host keeps every marker decoder unchanged; the only host change is that the ECS copy loop
// (main.go:993-1021) must hand over the RAW stuffed bytes (FF00 and RSTn left in place) because the
// reference's loop throws the restart markers away (SURVEY F3).  See INTEGRATION.md for the patch.
//
// NOT COMPILED IN THIS REPOSITORY'S IMAGE: there is no Go toolchain here (`go version`: command not
// found).  The C side of every call below is exercised by tests/ through ctypes with the same
// argument marshalling.
//
// Build:  CGO_ENABLED=1 CGO_CFLAGS="-I<repo>/include" CGO_LDFLAGS="-L<repo>/jpeg-decoder_amd -ljpeghip" go build .
package main

/*
#include <stdlib.h>
#include <string.h>
#include "jpeghip.h"
*/
import "C"

import (
	"fmt"
	"os"
	"unsafe"
)

var jdReady bool

// jdInit binds the library to every visible MI355X once (jd_init is idempotent).
func jdInit() {
	if jdReady {
		return
	}
	if rc := C.jd_init(nil, 0); rc != 0 {
		fmt.Printf("Error! %s\n", C.GoString(C.jd_strerror(rc)))
		os.Exit(1)
	}
	jdReady = true
}

// frameFromHeader marshals what Header holds at cut 1 (main.go:1467-1491) into a jd_frame.
func frameFromHeader(header *Header) C.jd_frame {
	var f C.jd_frame
	f.width = C.uint16_t(header.width)
	f.height = C.uint16_t(header.height)
	f.ncomp = C.uint8_t(len(header.cComponents))
	f.restart_interval = C.uint16_t(header.restartInterval)
	for i := range header.cComponents {
		c := &header.cComponents[i]
		f.comp[i].h = C.uint8_t(c.hSamplingFactor)
		f.comp[i].v = C.uint8_t(c.vSamplingFactor)
		f.comp[i].tq = C.uint8_t(c.qTableId)
		f.comp[i].td = C.uint8_t(c.dcHuffmanTableId)
		f.comp[i].ta = C.uint8_t(c.acHuffmanTableId)
	}
	for i := range header.qTables { // natural order bytes, exactly as main.go:61-75 stores them
		q := &header.qTables[i]
		for k := 0; k < 64; k++ {
			f.qt[q.Id][k] = C.uint8_t(q.table[k])
		}
		f.qt_present |= 1 << uint(q.Id)
	}
	for i := range header.huffmanTables { // codesOfLen + symbols (main.go:853-870); codes[] is not needed
		t := &header.huffmanTables[i]
		cls := 1
		if t.dc {
			cls = 0
		}
		h := &f.huff[cls][t.Id]
		for k := 0; k < 16; k++ {
			h.bits[k] = C.uint8_t(t.codesOfLen[k])
		}
		for k := range t.symbols {
			h.vals[k] = C.uint8_t(t.symbols[k])
		}
		h.nvals = C.uint16_t(len(t.symbols))
		h.present = 1
	}
	return f
}

// decodeOnGPU replaces main.go:1022-1032 and 1054-1057.  rawECS is the stuffed entropy-coded segment
// between the SOS header and EOI.  The result is H x W x 3 bytes, top-down, R,G,B.
func decodeOnGPU(header *Header, rawECS []byte) []byte {
	jdInit()
	f := frameFromHeader(header)
	if rc := C.jd_frame_check(&f); rc != 0 {
		fmt.Printf("%s\n", C.GoString(C.jd_strerror(rc)))
		os.Exit(1)
	}
	rgb := make([]byte, header.width*header.height*3)
	// cgo rule: C must not keep Go pointers after the call returns -- jd_decode is synchronous and
	// copies the ECS into device memory before it returns.
	rc := C.jd_decode(&f, (*C.uint8_t)(unsafe.Pointer(&rawECS[0])), C.size_t(len(rawECS)), 1, nil, 0,
		(*C.uint8_t)(unsafe.Pointer(&rgb[0])), C.size_t(header.width*3), 0)
	if rc != 0 {
		// the reference prints and exits 1 on every decode error (main.go:189-192, 1014-1016)
		fmt.Printf("%s\n", C.GoString(C.jd_strerror(rc)))
		os.Exit(1)
	}
	return rgb
}

// blocksFromRGB refills Header.blocks so the UNMODIFIED writeBitMap (main.go:1171-1218) can be kept:
// ch1=R, ch2=G, ch3=B at block (x/8, y/8), element (x%8)+(y%8)*8 (main.go:1199-1211).
func blocksFromRGB(header *Header, rgb []byte) {
	blocks := *header.blocks
	for y := 0; y < header.height; y++ {
		for x := 0; x < header.width; x++ {
			b := &blocks[(x/8)+(y/8)*header.blockWidthReal]
			pi := (x % 8) + (y%8)*8
			o := (y*header.width + x) * 3
			b.ch1[pi], b.ch2[pi], b.ch3[pi] = int(rgb[o]), int(rgb[o+1]), int(rgb[o+2])
		}
	}
}

// writeBitMapRGB is the buffered form of writeBitMap fed straight from RGB8 (SURVEY row N1): the
// library lays the file out byte-for-byte as main.go:1171-1218 does.
func writeBitMapRGB(name string, rgb []byte, width, height int) {
	n := C.jd_bmp_encode(nil, 0, C.int(width), C.int(height), nil, 0)
	out := make([]byte, int(n))
	C.jd_bmp_encode((*C.uint8_t)(unsafe.Pointer(&rgb[0])), C.size_t(width*3), C.int(width), C.int(height),
		(*C.uint8_t)(unsafe.Pointer(&out[0])), C.size_t(n))
	if err := os.WriteFile(name, out, 0o644); err != nil {
		fmt.Printf("Error! %v\n", err)
		os.Exit(1)
	}
}

// ---- throughput form (SURVEY row N3): main's file loop (main.go:1509-1512) over pinned-I/O batches ----------------
// A gpuSlot is one JD_BATCH_PINNED_IO batch: the library owns pinned staging for the entropy-coded bytes of all its
// images and for all their RGB.  The host keeps two or three slots going: while one is on the device (H2D copy ->
// kernels -> D2H copy, all asynchronous) it fills the next.  All memory behind these pointers is C memory, so Go may
// keep them between calls.
type gpuSlot struct {
	b       *C.jd_batch
	names   []string
	frames  []C.jd_frame
	pending bool
}

// newSlot plans a slot for images with these frames (same geometry and tables as the files that will be staged).
func newSlot(frames []C.jd_frame, ecsCap int) *gpuSlot {
	jdInit()
	caps := make([]C.size_t, len(frames))
	for i := range caps {
		caps[i] = C.size_t(ecsCap)
	}
	s := &gpuSlot{frames: frames, names: make([]string, len(frames))}
	if rc := C.jd_batch_create(0, C.int(len(frames)), &frames[0], &caps[0], C.JD_BATCH_PINNED_IO, &s.b); rc != 0 {
		fmt.Printf("%s\n", C.GoString(C.jd_strerror(rc)))
		os.Exit(1)
	}
	return s
}

// stage copies image i's raw (stuffed) entropy-coded segment into the slot's pinned staging.
func (s *gpuSlot) stage(i int, name string, rawECS []byte) {
	var dst *C.uint8_t
	if rc := C.jd_batch_stage_reserve(s.b, C.int(i), C.size_t(len(rawECS)), &dst); rc != 0 {
		fmt.Printf("%s\n", C.GoString(C.jd_strerror(rc)))
		os.Exit(1)
	}
	copy(unsafe.Slice((*byte)(unsafe.Pointer(dst)), len(rawECS)), rawECS)
	s.names[i] = name
}

func (s *gpuSlot) submit() {
	if rc := C.jd_batch_submit(s.b, nil); rc != 0 {
		fmt.Printf("%s\n", C.GoString(C.jd_strerror(rc)))
		os.Exit(1)
	}
	s.pending = true
}

// finish waits for the slot and writes its BMPs straight from the pinned RGB staging.
func (s *gpuSlot) finish() {
	if !s.pending {
		return
	}
	if rc := C.jd_batch_wait(s.b); rc != 0 {
		fmt.Printf("%s\n", C.GoString(C.jd_strerror(rc)))
		os.Exit(1)
	}
	for i := range s.frames {
		var p *C.uint8_t
		var stride C.size_t
		C.jd_batch_host_rgb(s.b, C.int(i), &p, &stride)
		w, h := int(s.frames[i].width), int(s.frames[i].height)
		writeBitMapRGB(s.names[i], unsafe.Slice((*byte)(unsafe.Pointer(p)), w*h*3), w, h)
	}
	s.pending = false
}

// ---- progressive files (SURVEY row N4) ------------------------------------------------------------------------
// decodeStartOfScan (main.go:942-1063) keeps parsing SOS / DHT / DRI between the scans of an SOF2 file; instead of
// running decodeHuffmanData per scan (main.go:1032) it appends scanFromHeader(...) to a list, and at EOI
// (main.go:1053) hands the list to decodeScansOnGPU.  `file` is the whole file in memory (os.ReadFile), off / length
// locate the scan's raw entropy-coded bytes in it.
func scanFromHeader(header *Header, off, length int) C.jd_scan {
	var sc C.jd_scan
	sc.progressive = 1
	sc.ss = C.uint8_t(header.startOfSelection)
	sc.se = C.uint8_t(header.endOfSelection)
	sc.ah = C.uint8_t(header.successiveApproximationHigh)
	sc.al = C.uint8_t(header.successiveApproximationLow)
	sc.ecs_off = C.uint64_t(off)
	sc.ecs_len = C.uint64_t(length)
	n := 0
	for i := range header.cComponents {
		c := &header.cComponents[i]
		if c.usedInScan {
			sc.comp[n] = C.uint8_t(i)
			sc.td[n] = C.uint8_t(c.dcHuffmanTableId)
			sc.ta[n] = C.uint8_t(c.acHuffmanTableId)
			n++
		}
	}
	sc.ncomp = C.uint8_t(n)
	f := frameFromHeader(header) // snapshot of header.huffmanTables as they stand at this scan
	sc.huff = f.huff
	return sc
}

func decodeScansOnGPU(header *Header, scans []C.jd_scan, file []byte) []byte {
	jdInit()
	f := frameFromHeader(header)
	rgb := make([]byte, header.width*header.height*3)
	rc := C.jd_decode_scans(&f, &scans[0], C.int(len(scans)), (*C.uint8_t)(unsafe.Pointer(&file[0])), C.size_t(len(file)),
		(*C.uint8_t)(unsafe.Pointer(&rgb[0])), C.size_t(header.width*3), 0)
	if rc != 0 {
		fmt.Printf("%s\n", C.GoString(C.jd_strerror(rc)))
		os.Exit(1)
	}
	return rgb
}

// Restart-interval-range sharding (SURVEY 8e(ii)): one DRI picture over several GPUs.  Each band of jd_split_bands is an
// image of its own; device k decodes band k (one goroutine / OS thread per device) straight into its rows of the picture.
func decodeBandsOnGPUs(header *Header, rawECS []byte, devices int) []byte {
	jdInit()
	// band k goes to device k: never ask for more bands than jd_init bound devices
	if have := int(C.jd_device_count()); devices > have {
		devices = have
	}
	if devices < 1 {
		devices = 1
	}
	f := frameFromHeader(header)
	bands := make([]C.jd_band, devices)
	var n C.int
	if rc := C.jd_split_bands(&f, (*C.uint8_t)(unsafe.Pointer(&rawECS[0])), C.size_t(len(rawECS)), C.int(devices), &bands[0], &n); rc != 0 {
		return decodeOnGPU(header, rawECS) // no restart markers at the MCU-row starts: the whole picture on one device
	}
	stride := int(header.width) * 3
	rgb := make([]byte, stride*int(header.height))
	done := make(chan C.int, int(n))
	for k := 0; k < int(n); k++ {
		go func(k int) {
			b := &bands[k]
			seg := rawECS[int(b.ecs_off) : int(b.ecs_off)+int(b.ecs_len)]
			rows := rgb[int(b.y0)*stride:]
			done <- C.jd_decode(&b.frame, (*C.uint8_t)(unsafe.Pointer(&seg[0])), C.size_t(len(seg)), 1, nil, 0,
				(*C.uint8_t)(unsafe.Pointer(&rows[0])), C.size_t(stride), C.int(k))
		}(k)
	}
	for k := 0; k < int(n); k++ {
		if rc := <-done; rc != 0 {
			fmt.Printf("%s\n", C.GoString(C.jd_strerror(rc)))
			os.Exit(1)
		}
	}
	return rgb
}
