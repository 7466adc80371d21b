// RUN: neptune-opt %s --neptuneir-to-llvm
// 3-D 7-point Laplacian, 512^3 f64: lap = dxinv2 * ((xm + xp + ym + yp + zm + zp) - 6 c)
// Authored for the MI355X backend: the reference README cites a file of this name
// (README.md:60-64) but ships none.  Structure follows the reference's own smoke inputs
// (test/smoke_tests/smoke_time_advance.mlir:3-31, 53-59, 82-84): interior bounds,
// region signature (index x rank, temp x inputs), body ops in a fixed textual order.

#loc = #neptune_ir.location<"cell">
#b   = #neptune_ir.bounds<lb = [0, 0, 0], ub = [512, 512, 512]>

!temp  = !neptune_ir.temp<element = f64, bounds = #b, location = #loc>
!field = !neptune_ir.field<element = f64, bounds = #b, location = #loc>

module {
  neptune_ir.linear_opdef @lap3d : (!temp) -> !temp {
  ^bb0(%u: !temp):
    %r = neptune_ir.apply(%u) attributes {bounds = #neptune_ir.bounds<lb = [1, 1, 1], ub = [511, 511, 511]>}
      : (!temp) -> !temp {
      ^bb0(%i0: index, %i1: index, %i2: index, %u_in: !temp):
        %c = neptune_ir.access %u_in[0, 0, 0] : !temp -> f64
        %xm = neptune_ir.access %u_in[-1, 0, 0] : !temp -> f64
        %xp = neptune_ir.access %u_in[1, 0, 0] : !temp -> f64
        %ym = neptune_ir.access %u_in[0, -1, 0] : !temp -> f64
        %yp = neptune_ir.access %u_in[0, 1, 0] : !temp -> f64
        %zm = neptune_ir.access %u_in[0, 0, -1] : !temp -> f64
        %zp = neptune_ir.access %u_in[0, 0, 1] : !temp -> f64

        %six    = arith.constant 6.0 : f64
        %dxinv2 = arith.constant 0.0625 : f64
        %t0  = arith.addf %xm, %xp : f64
        %t1  = arith.addf %t0, %ym : f64
        %t2  = arith.addf %t1, %yp : f64
        %t3  = arith.addf %t2, %zm : f64
        %t4  = arith.addf %t3, %zp : f64
        %t5  = arith.mulf %six, %c : f64
        %t6  = arith.subf %t4, %t5 : f64
        %lap = arith.mulf %dxinv2, %t6 : f64
        neptune_ir.yield %lap : f64
      }
    neptune_ir.return %r : !temp
  }

  func.func @entry(%out: memref<?x?x?xf64>, %in: memref<?x?x?xf64>) -> memref<?x?x?xf64> {
    %fout = neptune_ir.wrap %out : memref<?x?x?xf64> -> !field
    %fin  = neptune_ir.wrap %in  : memref<?x?x?xf64> -> !field
    %u0   = neptune_ir.load %fin : !field -> !temp
    %y    = neptune_ir.apply_linear @lap3d(%u0) : (!temp) -> !temp
    neptune_ir.store %y to %fout : !temp to !field
    %res  = neptune_ir.unwrap %fout : !field -> memref<?x?x?xf64>
    func.return %res : memref<?x?x?xf64>
  }
}
