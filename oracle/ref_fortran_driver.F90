! ref_fortran_driver.F90 -- drives the REFERENCE's Fortran compute_and_apply_rhs
! (compute_and_apply_rhs_test/fortran/routine_mod.F90:7) on arrays read from a
! binary file and writes every mutated array back.
!
! TEST INFRASTRUCTURE ONLY.  oracle/Makefile compiles this file together with the
! reference's Fortran modules *where they lie* under /root/reference (kinds,
! element_state_mod, element_mod, physical_constants, derivative_mod_base,
! hybvcoord_mod, routine_mod, ...) into oracle/_ref/fortran_driver.  Nothing of
! the reference is copied here: the program only "use"s its modules.
!
! File format (little-endian stream, written/read by tests/golden/make_golden.py):
!   int32  nelem, n0, np1, nm1, qn0      (0-based time levels; qn0=-1 -> dry)
!   real64 dt2, eta_ave_w, ps0, hyai(nlev+1), Dvv(np*np) [C order Dvv[i][j]]
!   then the 16 arrays of Homme::Arrays (cxx/pointers_only/data_structures.hpp:18-44)
!   in that order and in the C++ element-major layout.
! Output: state_dp3d, state_v, state_T, eta_dot_dpdn, omega_p, phi, vn0 (C++ layout).
! Optional third argument: a second output file that receives the same seven arrays plus
! Qdp, D, fcor in the reference's NATIVE Fortran array-element order, element by element
! (write(u) elem(ie)%state%v ...), i.e. the flat arrays v(np,np,2,nlev,timelevels,nelemd)
! etc. a Fortran host would hand over — the pin for the layout kernels (csrc/caar_layout.hip).
!
! Index map (SURVEY.md 8a): C++ [ie][..][a][b] == Fortran elem(ie+1)%..(a+1,b+1,..).
program ref_fortran_driver
  use kinds
  use element_state_mod
  use element_mod
  use derivative_mod_base
  use hybvcoord_mod
  use routine_mod, only : compute_and_apply_rhs
  implicit none

  type (element_t), allocatable :: elem(:)
  type (derivative_t) :: deriv
  type (hvcoord_t)    :: hvcoord
  integer(kind=4) :: hdr(5)
  integer :: ne, c_n0, c_np1, c_nm1, c_qn0, f_qn0
  real (kind=real_kind) :: dt2, eta_ave_w
  real (kind=real_kind) :: dvv_c(np*np)
  real (kind=real_kind), allocatable :: b(:)
  integer :: ie, a, bb, c, r, k, t, q, u
  integer(kind=8) :: o
  character(len=512) :: fin, fout, fnative

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  fnative = ''
  if (command_argument_count() >= 3) call get_command_argument(3, fnative)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) hdr
  ne = hdr(1); c_n0 = hdr(2); c_np1 = hdr(3); c_nm1 = hdr(4); c_qn0 = hdr(5)
  nelemd = ne
  read(u) dt2, eta_ave_w, hvcoord%ps0
  read(u) hvcoord%hyai
  read(u) dvv_c
  do a = 1, np
    do bb = 1, np
      deriv%Dvv(a, bb) = dvv_c((a-1)*np + bb)     ! C Dvv[i][j] == Fortran Dvv(i+1,j+1)
    end do
  end do
  allocate(elem(ne))

  ! --- elem_D, elem_Dinv  [ie][a][b][r][c]
  allocate(b(ne*np*np*4))
  read(u) b
  do ie = 1, ne; do a = 1, np; do bb = 1, np; do r = 1, 2; do c = 1, 2
    elem(ie)%D(a,bb,r,c) = b(((((ie-1)*np + a-1)*np + bb-1)*2 + r-1)*2 + c)
  end do; end do; end do; end do; end do
  read(u) b
  do ie = 1, ne; do a = 1, np; do bb = 1, np; do r = 1, 2; do c = 1, 2
    elem(ie)%Dinv(a,bb,r,c) = b(((((ie-1)*np + a-1)*np + bb-1)*2 + r-1)*2 + c)
  end do; end do; end do; end do; end do
  deallocate(b)

  ! --- fcor, spheremp, metdet, rmetdet  [ie][a][b]
  allocate(b(ne*np*np))
  read(u) b
  do ie = 1, ne; do a = 1, np; do bb = 1, np
    elem(ie)%fcor(a,bb) = b(((ie-1)*np + a-1)*np + bb)
  end do; end do; end do
  read(u) b
  do ie = 1, ne; do a = 1, np; do bb = 1, np
    elem(ie)%spheremp(a,bb) = b(((ie-1)*np + a-1)*np + bb)
  end do; end do; end do
  read(u) b
  do ie = 1, ne; do a = 1, np; do bb = 1, np
    elem(ie)%metdet(a,bb) = b(((ie-1)*np + a-1)*np + bb)
  end do; end do; end do
  read(u) b
  do ie = 1, ne; do a = 1, np; do bb = 1, np
    elem(ie)%rmetdet(a,bb) = b(((ie-1)*np + a-1)*np + bb)
  end do; end do; end do
  deallocate(b)

  ! --- state_dp3d [ie][t][k][a][b]
  allocate(b(ne*timelevels*nlev*np*np))
  read(u) b
  do ie = 1, ne; do t = 1, timelevels; do k = 1, nlev; do a = 1, np; do bb = 1, np
    elem(ie)%state%dp3d(a,bb,k,t) = b(((((ie-1)*timelevels + t-1)*nlev + k-1)*np + a-1)*np + bb)
  end do; end do; end do; end do; end do
  deallocate(b)
  ! --- state_v [ie][t][k][a][b][c]
  allocate(b(ne*timelevels*nlev*np*np*2))
  read(u) b
  do ie = 1, ne; do t = 1, timelevels; do k = 1, nlev; do a = 1, np; do bb = 1, np; do c = 1, 2
    elem(ie)%state%v(a,bb,c,k,t) = b((((((ie-1)*timelevels + t-1)*nlev + k-1)*np + a-1)*np + bb-1)*2 + c)
  end do; end do; end do; end do; end do; end do
  deallocate(b)
  ! --- state_T
  allocate(b(ne*timelevels*nlev*np*np))
  read(u) b
  do ie = 1, ne; do t = 1, timelevels; do k = 1, nlev; do a = 1, np; do bb = 1, np
    elem(ie)%state%T(a,bb,k,t) = b(((((ie-1)*timelevels + t-1)*nlev + k-1)*np + a-1)*np + bb)
  end do; end do; end do; end do; end do
  deallocate(b)
  ! --- state_phis [ie][a][b]
  allocate(b(ne*np*np))
  read(u) b
  do ie = 1, ne; do a = 1, np; do bb = 1, np
    elem(ie)%state%phis(a,bb) = b(((ie-1)*np + a-1)*np + bb)
  end do; end do; end do
  deallocate(b)
  ! --- state_Qdp [ie][q][2][k][a][b]  (Fortran Qdp(np,np,nlev,qsize_d,timelevels): slots 1..2 used)
  allocate(b(ne*qsize_d*2*nlev*np*np))
  read(u) b
  do ie = 1, ne; do q = 1, qsize_d; do t = 1, 2; do k = 1, nlev; do a = 1, np; do bb = 1, np
    elem(ie)%state%Qdp(a,bb,k,q,t) = b((((((ie-1)*qsize_d + q-1)*2 + t-1)*nlev + k-1)*np + a-1)*np + bb)
  end do; end do; end do; end do; end do; end do
  deallocate(b)
  ! --- derived_eta_dot_dpdn [ie][k<=nlev][a][b]
  allocate(b(ne*(nlev+1)*np*np))
  read(u) b
  do ie = 1, ne; do k = 1, nlev+1; do a = 1, np; do bb = 1, np
    elem(ie)%derived%eta_dot_dpdn(a,bb,k) = b((((ie-1)*(nlev+1) + k-1)*np + a-1)*np + bb)
  end do; end do; end do; end do
  deallocate(b)
  ! --- derived_omega_p, phi, pecnd [ie][k][a][b]
  allocate(b(ne*nlev*np*np))
  read(u) b
  do ie = 1, ne; do k = 1, nlev; do a = 1, np; do bb = 1, np
    elem(ie)%derived%omega_p(a,bb,k) = b((((ie-1)*nlev + k-1)*np + a-1)*np + bb)
  end do; end do; end do; end do
  read(u) b
  do ie = 1, ne; do k = 1, nlev; do a = 1, np; do bb = 1, np
    elem(ie)%derived%phi(a,bb,k) = b((((ie-1)*nlev + k-1)*np + a-1)*np + bb)
  end do; end do; end do; end do
  read(u) b
  do ie = 1, ne; do k = 1, nlev; do a = 1, np; do bb = 1, np
    elem(ie)%derived%pecnd(a,bb,k) = b((((ie-1)*nlev + k-1)*np + a-1)*np + bb)
  end do; end do; end do; end do
  deallocate(b)
  ! --- derived_vn0 [ie][k][a][b][c]
  allocate(b(ne*nlev*np*np*2))
  read(u) b
  do ie = 1, ne; do k = 1, nlev; do a = 1, np; do bb = 1, np; do c = 1, 2
    elem(ie)%derived%vn0(a,bb,c,k) = b(((((ie-1)*nlev + k-1)*np + a-1)*np + bb-1)*2 + c)
  end do; end do; end do; end do; end do
  deallocate(b)
  close(u)

  f_qn0 = c_qn0 + 1
  if (c_qn0 == -1) f_qn0 = -1
  call compute_and_apply_rhs(c_np1+1, c_nm1+1, c_n0+1, f_qn0, dt2, elem, hvcoord, deriv, 1, ne, eta_ave_w)

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  allocate(b(ne*timelevels*nlev*np*np))
  do ie = 1, ne; do t = 1, timelevels; do k = 1, nlev; do a = 1, np; do bb = 1, np
    b(((((ie-1)*timelevels + t-1)*nlev + k-1)*np + a-1)*np + bb) = elem(ie)%state%dp3d(a,bb,k,t)
  end do; end do; end do; end do; end do
  write(u) b
  deallocate(b)
  allocate(b(ne*timelevels*nlev*np*np*2))
  do ie = 1, ne; do t = 1, timelevels; do k = 1, nlev; do a = 1, np; do bb = 1, np; do c = 1, 2
    b((((((ie-1)*timelevels + t-1)*nlev + k-1)*np + a-1)*np + bb-1)*2 + c) = elem(ie)%state%v(a,bb,c,k,t)
  end do; end do; end do; end do; end do; end do
  write(u) b
  deallocate(b)
  allocate(b(ne*timelevels*nlev*np*np))
  do ie = 1, ne; do t = 1, timelevels; do k = 1, nlev; do a = 1, np; do bb = 1, np
    b(((((ie-1)*timelevels + t-1)*nlev + k-1)*np + a-1)*np + bb) = elem(ie)%state%T(a,bb,k,t)
  end do; end do; end do; end do; end do
  write(u) b
  deallocate(b)
  allocate(b(ne*(nlev+1)*np*np))
  do ie = 1, ne; do k = 1, nlev+1; do a = 1, np; do bb = 1, np
    b((((ie-1)*(nlev+1) + k-1)*np + a-1)*np + bb) = elem(ie)%derived%eta_dot_dpdn(a,bb,k)
  end do; end do; end do; end do
  write(u) b
  deallocate(b)
  allocate(b(ne*nlev*np*np))
  do ie = 1, ne; do k = 1, nlev; do a = 1, np; do bb = 1, np
    b((((ie-1)*nlev + k-1)*np + a-1)*np + bb) = elem(ie)%derived%omega_p(a,bb,k)
  end do; end do; end do; end do
  write(u) b
  do ie = 1, ne; do k = 1, nlev; do a = 1, np; do bb = 1, np
    b((((ie-1)*nlev + k-1)*np + a-1)*np + bb) = elem(ie)%derived%phi(a,bb,k)
  end do; end do; end do; end do
  write(u) b
  deallocate(b)
  allocate(b(ne*nlev*np*np*2))
  do ie = 1, ne; do k = 1, nlev; do a = 1, np; do bb = 1, np; do c = 1, 2
    b(((((ie-1)*nlev + k-1)*np + a-1)*np + bb-1)*2 + c) = elem(ie)%derived%vn0(a,bb,c,k)
  end do; end do; end do; end do; end do
  write(u) b
  deallocate(b)
  close(u)
  if (len_trim(fnative) > 0) then
    open(newunit=u, file=trim(fnative), access='stream', form='unformatted', status='replace')
    do ie = 1, ne
      write(u) elem(ie)%state%dp3d
    end do
    do ie = 1, ne
      write(u) elem(ie)%state%v
    end do
    do ie = 1, ne
      write(u) elem(ie)%state%T
    end do
    do ie = 1, ne
      write(u) elem(ie)%derived%eta_dot_dpdn
    end do
    do ie = 1, ne
      write(u) elem(ie)%derived%omega_p
    end do
    do ie = 1, ne
      write(u) elem(ie)%derived%phi
    end do
    do ie = 1, ne
      write(u) elem(ie)%derived%vn0
    end do
    do ie = 1, ne
      write(u) elem(ie)%state%Qdp(:,:,:,:,1:2)
    end do
    do ie = 1, ne
      write(u) elem(ie)%D
    end do
    do ie = 1, ne
      write(u) elem(ie)%fcor
    end do
    close(u)
  end if
end program ref_fortran_driver
