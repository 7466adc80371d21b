// RUN: neptune-opt %s --neptuneir-to-llvm
// 2-D 5-point Laplacian, 1024x1024 f64: lap = dxinv2 * ((n + s + w + e) - 4 c)
// Authored for the MI355X backend: the reference README cites a file of this name
// (README.md:60-64) but ships none.  Structure follows the reference's own smoke inputs
// (test/smoke_tests/smoke_time_advance.mlir:3-31, 53-59, 82-84): interior bounds,
// region signature (index x rank, temp x inputs), body ops in a fixed textual order.

#loc = #neptune_ir.location<"cell">
#b   = #neptune_ir.bounds<lb = [0, 0], ub = [1024, 1024]>

!temp  = !neptune_ir.temp<element = f64, bounds = #b, location = #loc>
!field = !neptune_ir.field<element = f64, bounds = #b, location = #loc>

module {
  neptune_ir.linear_opdef @lap2d : (!temp) -> !temp {
  ^bb0(%u: !temp):
    %r = neptune_ir.apply(%u) attributes {bounds = #neptune_ir.bounds<lb = [1, 1], ub = [1023, 1023]>}
      : (!temp) -> !temp {
      ^bb0(%i0: index, %i1: index, %u_in: !temp):
        %c = neptune_ir.access %u_in[0, 0] : !temp -> f64
        %n = neptune_ir.access %u_in[-1, 0] : !temp -> f64
        %s = neptune_ir.access %u_in[1, 0] : !temp -> f64
        %w = neptune_ir.access %u_in[0, -1] : !temp -> f64
        %e = neptune_ir.access %u_in[0, 1] : !temp -> f64

        %four   = arith.constant 4.0 : f64
        %dxinv2 = arith.constant 0.125 : f64
        %t0  = arith.addf %n, %s : f64
        %t1  = arith.addf %t0, %w : f64
        %t2  = arith.addf %t1, %e : f64
        %t3  = arith.mulf %four, %c : f64
        %t4  = arith.subf %t2, %t3 : f64
        %lap = arith.mulf %dxinv2, %t4 : f64
        neptune_ir.yield %lap : f64
      }
    neptune_ir.return %r : !temp
  }

  func.func @entry(%out: memref<?x?xf64>, %in: memref<?x?xf64>) -> memref<?x?xf64> {
    %fout = neptune_ir.wrap %out : memref<?x?xf64> -> !field
    %fin  = neptune_ir.wrap %in  : memref<?x?xf64> -> !field
    %u0   = neptune_ir.load %fin : !field -> !temp
    %y    = neptune_ir.apply_linear @lap2d(%u0) : (!temp) -> !temp
    neptune_ir.store %y to %fout : !temp to !field
    %res  = neptune_ir.unwrap %fout : !field -> memref<?x?xf64>
    func.return %res : memref<?x?xf64>
  }
}
