// RUN: neptune-opt %s --neptuneir-to-llvm
// 3-D 27-point stencil, 512^3 f32: lap = dxinv2 * (sum of the 26 neighbours (dim-0-major order) - 26 c)
// Authored for the MI355X backend: the reference README cites a file of this name
// (README.md:60-64) but ships none.  Structure follows the reference's own smoke inputs
// (test/smoke_tests/smoke_time_advance.mlir:3-31, 53-59, 82-84): interior bounds,
// region signature (index x rank, temp x inputs), body ops in a fixed textual order.

#loc = #neptune_ir.location<"cell">
#b   = #neptune_ir.bounds<lb = [0, 0, 0], ub = [512, 512, 512]>

!temp  = !neptune_ir.temp<element = f32, bounds = #b, location = #loc>
!field = !neptune_ir.field<element = f32, bounds = #b, location = #loc>

module {
  neptune_ir.linear_opdef @lap27 : (!temp) -> !temp {
  ^bb0(%u: !temp):
    %r = neptune_ir.apply(%u) attributes {bounds = #neptune_ir.bounds<lb = [1, 1, 1], ub = [511, 511, 511]>}
      : (!temp) -> !temp {
      ^bb0(%i0: index, %i1: index, %i2: index, %u_in: !temp):
        %c = neptune_ir.access %u_in[0, 0, 0] : !temp -> f32
        %ammm = neptune_ir.access %u_in[-1, -1, -1] : !temp -> f32
        %ammz = neptune_ir.access %u_in[-1, -1, 0] : !temp -> f32
        %ammp = neptune_ir.access %u_in[-1, -1, 1] : !temp -> f32
        %amzm = neptune_ir.access %u_in[-1, 0, -1] : !temp -> f32
        %amzz = neptune_ir.access %u_in[-1, 0, 0] : !temp -> f32
        %amzp = neptune_ir.access %u_in[-1, 0, 1] : !temp -> f32
        %ampm = neptune_ir.access %u_in[-1, 1, -1] : !temp -> f32
        %ampz = neptune_ir.access %u_in[-1, 1, 0] : !temp -> f32
        %ampp = neptune_ir.access %u_in[-1, 1, 1] : !temp -> f32
        %azmm = neptune_ir.access %u_in[0, -1, -1] : !temp -> f32
        %azmz = neptune_ir.access %u_in[0, -1, 0] : !temp -> f32
        %azmp = neptune_ir.access %u_in[0, -1, 1] : !temp -> f32
        %azzm = neptune_ir.access %u_in[0, 0, -1] : !temp -> f32
        %azzp = neptune_ir.access %u_in[0, 0, 1] : !temp -> f32
        %azpm = neptune_ir.access %u_in[0, 1, -1] : !temp -> f32
        %azpz = neptune_ir.access %u_in[0, 1, 0] : !temp -> f32
        %azpp = neptune_ir.access %u_in[0, 1, 1] : !temp -> f32
        %apmm = neptune_ir.access %u_in[1, -1, -1] : !temp -> f32
        %apmz = neptune_ir.access %u_in[1, -1, 0] : !temp -> f32
        %apmp = neptune_ir.access %u_in[1, -1, 1] : !temp -> f32
        %apzm = neptune_ir.access %u_in[1, 0, -1] : !temp -> f32
        %apzz = neptune_ir.access %u_in[1, 0, 0] : !temp -> f32
        %apzp = neptune_ir.access %u_in[1, 0, 1] : !temp -> f32
        %appm = neptune_ir.access %u_in[1, 1, -1] : !temp -> f32
        %appz = neptune_ir.access %u_in[1, 1, 0] : !temp -> f32
        %appp = neptune_ir.access %u_in[1, 1, 1] : !temp -> f32

        %c26    = arith.constant 26.0 : f32
        %dxinv2 = arith.constant 0.015625 : f32
        %s0 = arith.addf %ammm, %ammz : f32
        %s1 = arith.addf %s0, %ammp : f32
        %s2 = arith.addf %s1, %amzm : f32
        %s3 = arith.addf %s2, %amzz : f32
        %s4 = arith.addf %s3, %amzp : f32
        %s5 = arith.addf %s4, %ampm : f32
        %s6 = arith.addf %s5, %ampz : f32
        %s7 = arith.addf %s6, %ampp : f32
        %s8 = arith.addf %s7, %azmm : f32
        %s9 = arith.addf %s8, %azmz : f32
        %s10 = arith.addf %s9, %azmp : f32
        %s11 = arith.addf %s10, %azzm : f32
        %s12 = arith.addf %s11, %azzp : f32
        %s13 = arith.addf %s12, %azpm : f32
        %s14 = arith.addf %s13, %azpz : f32
        %s15 = arith.addf %s14, %azpp : f32
        %s16 = arith.addf %s15, %apmm : f32
        %s17 = arith.addf %s16, %apmz : f32
        %s18 = arith.addf %s17, %apmp : f32
        %s19 = arith.addf %s18, %apzm : f32
        %s20 = arith.addf %s19, %apzz : f32
        %s21 = arith.addf %s20, %apzp : f32
        %s22 = arith.addf %s21, %appm : f32
        %s23 = arith.addf %s22, %appz : f32
        %s24 = arith.addf %s23, %appp : f32
        %t0  = arith.mulf %c26, %c : f32
        %t1  = arith.subf %s24, %t0 : f32
        %lap = arith.mulf %dxinv2, %t1 : f32
        neptune_ir.yield %lap : f32
      }
    neptune_ir.return %r : !temp
  }

  func.func @entry(%out: memref<?x?x?xf32>, %in: memref<?x?x?xf32>) -> memref<?x?x?xf32> {
    %fout = neptune_ir.wrap %out : memref<?x?x?xf32> -> !field
    %fin  = neptune_ir.wrap %in  : memref<?x?x?xf32> -> !field
    %u0   = neptune_ir.load %fin : !field -> !temp
    %y    = neptune_ir.apply_linear @lap27(%u0) : (!temp) -> !temp
    neptune_ir.store %y to %fout : !temp to !field
    %res  = neptune_ir.unwrap %fout : !field -> memref<?x?x?xf32>
    func.return %res : memref<?x?x?xf32>
  }
}
