// 3-D 13-point 4th-order Laplacian (radius-2 star), 20x18x256 f64:
// lap = dxinv2 * (sum_d (-(u[-2]+u[+2]) + 16 (u[-1]+u[+1])) - 90 c) / 12, ops in the textual order below.
// Authored for the MI355X backend: exercises the march kernel with 5 live planes and 2-deep halos.
#loc = #neptune_ir.location<"cell">
#b   = #neptune_ir.bounds<lb = [0, 0, 0], ub = [20, 18, 256]>
!temp  = !neptune_ir.temp<element = f64, bounds = #b, location = #loc>
!field = !neptune_ir.field<element = f64, bounds = #b, location = #loc>
module {
  neptune_ir.linear_opdef @lap13 : (!temp) -> !temp {
  ^bb0(%u: !temp):
    %r = neptune_ir.apply(%u) attributes {bounds = #neptune_ir.bounds<lb = [2, 2, 2], ub = [18, 16, 254]>}
      : (!temp) -> !temp {
      ^bb0(%i0: index, %i1: index, %i2: index, %u_in: !temp):
        %c = neptune_ir.access %u_in[0, 0, 0] : !temp -> f64
        %xm2 = neptune_ir.access %u_in[-2, 0, 0] : !temp -> f64
        %xm1 = neptune_ir.access %u_in[-1, 0, 0] : !temp -> f64
        %xp1 = neptune_ir.access %u_in[1, 0, 0] : !temp -> f64
        %xp2 = neptune_ir.access %u_in[2, 0, 0] : !temp -> f64
        %ym2 = neptune_ir.access %u_in[0, -2, 0] : !temp -> f64
        %ym1 = neptune_ir.access %u_in[0, -1, 0] : !temp -> f64
        %yp1 = neptune_ir.access %u_in[0, 1, 0] : !temp -> f64
        %yp2 = neptune_ir.access %u_in[0, 2, 0] : !temp -> f64
        %zm2 = neptune_ir.access %u_in[0, 0, -2] : !temp -> f64
        %zm1 = neptune_ir.access %u_in[0, 0, -1] : !temp -> f64
        %zp1 = neptune_ir.access %u_in[0, 0, 1] : !temp -> f64
        %zp2 = neptune_ir.access %u_in[0, 0, 2] : !temp -> f64
        %c16 = arith.constant 16.0 : f64
        %c90 = arith.constant 90.0 : f64
        %k12 = arith.constant 0.0078125 : f64
        %xo = arith.addf %xm2, %xp2 : f64
        %xi = arith.addf %xm1, %xp1 : f64
        %xs = arith.mulf %c16, %xi : f64
        %xt = arith.subf %xs, %xo : f64
        %yo = arith.addf %ym2, %yp2 : f64
        %yi = arith.addf %ym1, %yp1 : f64
        %ys = arith.mulf %c16, %yi : f64
        %yt = arith.subf %ys, %yo : f64
        %accy = arith.addf %xt, %yt : f64
        %zo = arith.addf %zm2, %zp2 : f64
        %zi = arith.addf %zm1, %zp1 : f64
        %zs = arith.mulf %c16, %zi : f64
        %zt = arith.subf %zs, %zo : f64
        %accz = arith.addf %accy, %zt : f64
        %cc = arith.mulf %c90, %c : f64
        %df = arith.subf %accz, %cc : f64
        %lap = arith.mulf %k12, %df : f64
        neptune_ir.yield %lap : f64
      }
    neptune_ir.return %r : !temp
  }
  func.func @entry(%out: memref<?x?x?xf64>, %in: memref<?x?x?xf64>) -> memref<?x?x?xf64> {
    %fout = neptune_ir.wrap %out : memref<?x?x?xf64> -> !field
    %fin  = neptune_ir.wrap %in  : memref<?x?x?xf64> -> !field
    %u0   = neptune_ir.load %fin : !field -> !temp
    %y    = neptune_ir.apply_linear @lap13(%u0) : (!temp) -> !temp
    neptune_ir.store %y to %fout : !temp to !field
    %res  = neptune_ir.unwrap %fout : !field -> memref<?x?x?xf64>
    func.return %res : memref<?x?x?xf64>
  }
}
