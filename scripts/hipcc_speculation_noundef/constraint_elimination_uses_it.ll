declare i32 @llvm.umin.i32(i32, i32)

define i32 @f(i32 %U, i32 %c, i1 %b) {
entry:
  %c3 = and i32 %c, 3
  %s = shl nuw nsw i32 %c3, 2
  %p = or disjoint i32 %s, 48
  %cmp = icmp ult i32 %p, %U
  %sub = sub nuw i32 %U, %p
  %m = call noundef i32 @llvm.umin.i32(i32 %sub, i32 4)
  %sel = select i1 %cmp, i32 %m, i32 0
  %um1 = add i32 %U, -1
  br i1 %b, label %then, label %exit

then:
  %g = icmp ugt i32 %um1, 43
  %r = select i1 %g, i32 %sel, i32 7
  ret i32 %r

exit:
  ret i32 %sel
}
