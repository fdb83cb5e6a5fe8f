! caar_f90_driver.F90 -- a Fortran host running compute_and_apply_rhs on the MI355X.
!
! Same flow and same printed lines as the reference's Fortran driver
! (compute_and_apply_rhs_test/fortran/main.F90: closed-form initialisation :103-154 with
! the single-precision Dvv literals :83-96, norms of the np1 state before :168-194 and
! after :278-304 the call), but the fields live in flat Fortran-ordered arrays and the
! call goes through caar_mod to libcaar_hip.so.  argv(1) = number of elements (default 3).
program caar_f90_driver
  use iso_c_binding
  use caar_mod
  implicit none
  integer, parameter :: np = 4, nlev = 72, qsize_d = 1, timelevels = 3
  integer :: nelemd = 3
  real(c_double), allocatable, target :: D(:,:,:,:,:), Dinv(:,:,:,:,:)
  real(c_double), allocatable, target :: fcor(:,:,:), spheremp(:,:,:), metdet(:,:,:), rmetdet(:,:,:), phis(:,:,:)
  real(c_double), allocatable, target :: dp3d(:,:,:,:,:), v(:,:,:,:,:,:), T(:,:,:,:,:), Qdp(:,:,:,:,:,:)
  real(c_double), allocatable, target :: eta_dot_dpdn(:,:,:,:), omega_p(:,:,:,:), phi(:,:,:,:), pecnd(:,:,:,:)
  real(c_double), allocatable, target :: vn0(:,:,:,:,:)
  real(c_double), target :: Dvv_c(np*np)
  real(c_double) :: Dvv(np,np), nrm(3)
  real(c_double) :: gi(np,np), gj(np,np), zk, ze
  real, parameter :: dvv_single(np*np) = (/ -3.0, -0.80901699437494745, 0.30901699437494745, -0.5, &
      4.0450849718747373, 0.0, -1.1180339887498949, 1.5450849718747370, &
      -1.5450849718747370, 1.1180339887498949, 0.0, -4.0450849718747373, &
      0.5, -0.30901699437494745, 0.80901699437494745, 3.0 /)
  type(caar_dims_t) :: dims
  type(caar_arrays_t) :: a
  type(caar_params_t) :: prm
  type(c_ptr) :: ctx
  integer :: i, j, k, ie, tl
  character(len=32) :: arg

  if (command_argument_count() >= 1) then
    call get_command_argument(1, arg)
    read (arg, *) nelemd
  end if
  print *, "Main: nelemd = ", nelemd
  if (caar_device_count() < 1) then
    print *, "No HIP device is visible: the MI355X path cannot run (there is no CPU fallback)."
    error stop 1
  end if

  allocate(D(np,np,2,2,nelemd), Dinv(np,np,2,2,nelemd))
  allocate(fcor(np,np,nelemd), spheremp(np,np,nelemd), metdet(np,np,nelemd), rmetdet(np,np,nelemd), phis(np,np,nelemd))
  allocate(dp3d(np,np,nlev,timelevels,nelemd), v(np,np,2,nlev,timelevels,nelemd), T(np,np,nlev,timelevels,nelemd))
  allocate(Qdp(np,np,nlev,qsize_d,2,nelemd))
  allocate(eta_dot_dpdn(np,np,nlev+1,nelemd), omega_p(np,np,nlev,nelemd), phi(np,np,nlev,nelemd), pecnd(np,np,nlev,nelemd))
  allocate(vn0(np,np,2,nlev,nelemd))

  ! Derivative matrix: the reference driver's values are DEFAULT-REAL (single precision) literals widened to
  ! double (main.F90:83-96) -- that is what its golden vectors were made with, so they stay single here.
  ! reshape fills column-major: Dvv(i,j) = value((j-1)*np + i); the C ABI wants row-major Dvv[i][j].
  Dvv = real(reshape(dvv_single, (/ np, np /)), c_double)
  Dvv_c = reshape(transpose(Dvv), (/ np*np /))

  ! Closed-form fields of the reference driver (main.F90:103-154), written as whole-level array expressions
  ! over the GLL point indices gi(i,j) = i, gj(i,j) = j; every expression keeps the reference's operand order
  ! (the state is compared with the reference's golden vectors digit for digit).
  gi = spread((/ (real(i, c_double), i = 1, np) /), dim=2, ncopies=np)
  gj = spread((/ (real(j, c_double), j = 1, np) /), dim=1, ncopies=np)
  D = 0
  Dinv = 0
  eta_dot_dpdn = 0
  Qdp = 0
  vn0 = 1.0
  pecnd = 1.0
  do ie = 1, nelemd
    ze = ie
    fcor(:,:,ie) = sin(gi + gj)
    metdet(:,:,ie) = gi*gj
    rmetdet(:,:,ie) = 1.0d0/metdet(:,:,ie)
    spheremp(:,:,ie) = 2*gi
    phis(:,:,ie) = gi + gj
    D(:,:,1,1,ie) = 1.0
    D(:,:,2,2,ie) = 2.0
    Dinv(:,:,1,1,ie) = 1.0
    Dinv(:,:,2,2,ie) = 0.5
    do k = 1, nlev
      zk = k
      phi(:,:,k,ie) = cos(gi + 3*gj) + zk
      omega_p(:,:,k,ie) = gj*gj
      Qdp(:,:,k,1,1,ie) = 1.0 + sin(gi*gj*zk)
      do tl = 1, timelevels
        dp3d(:,:,k,tl,ie) = 10*zk + ze + gi + gj + tl
        v(:,:,1,k,tl,ie) = 1.0 + zk/2 + gi + gj + ze/5 + tl*2.0
        v(:,:,2,k,tl,ie) = 1.0 + zk/2 + gi + gj + ze/5 + tl*3.0
        T(:,:,k,tl,ie) = 1000 - zk - gi - gj + ze/10 + tl
      end do
    end do
  end do

  dims%np = np; dims%nlev = nlev; dims%qsize_d = qsize_d; dims%timelevels = timelevels; dims%num_elems = nelemd
  a%elem_D = c_loc(D); a%elem_Dinv = c_loc(Dinv); a%elem_fcor = c_loc(fcor); a%elem_spheremp = c_loc(spheremp)
  a%elem_metdet = c_loc(metdet); a%elem_rmetdet = c_loc(rmetdet)
  a%elem_state_dp3d = c_loc(dp3d); a%elem_state_v = c_loc(v); a%elem_state_T = c_loc(T)
  a%elem_state_phis = c_loc(phis); a%elem_state_Qdp = c_loc(Qdp)
  a%elem_derived_eta_dot_dpdn = c_loc(eta_dot_dpdn); a%elem_derived_omega_p = c_loc(omega_p)
  a%elem_derived_phi = c_loc(phi); a%elem_derived_pecnd = c_loc(pecnd); a%elem_derived_vn0 = c_loc(vn0)

  ! the reference's 1-based (np1,nm1,n0,qn0) = (2,3,1,1) and elements nets..nete = 1..nelemd
  prm%nets = 0; prm%nete = nelemd; prm%n0 = 0; prm%np1 = 1; prm%nm1 = 2; prm%qn0 = 0
  prm%dt2 = 1.0d0; prm%eta_ave_w = 1.0d0
  prm%rrearth = 1.0d0/6.376d6; prm%Rwater_vapor = 461.5d0; prm%Rgas = 287.04d0; prm%kappa = 287.04d0/1005.0d0
  prm%ps0 = 10.0d0; prm%hyai0 = nlev + 1          ! hvcoord%hyai(1) = nlev + 2 - 1 (main.F90:160-162)
  prm%Dvv = c_loc(Dvv_c)

  call caar_check(caar_create(ctx, dims, 0_c_int), 'caar_create')
  call caar_check(caar_upload_f90(ctx, a, 0_c_int, nelemd), 'caar_upload_f90')
  call caar_check(caar_state_norms(ctx, prm%np1, 0_c_int, nelemd, nrm), 'caar_state_norms')
  print *, "||v||_2  = ", nrm(1)
  print *, "||T||_2  = ", nrm(2)
  print *, "||dp||_2 = ", nrm(3)
  print *, 'Main, np=', np

  call caar_check(caar_run(ctx, prm), 'caar_run')          ! == call compute_and_apply_rhs(...)
  call caar_check(caar_download_f90(ctx, a, 0_c_int, nelemd, 0_c_int), 'caar_download_f90')
  call caar_check(caar_sync(ctx), 'caar_sync')

  call caar_check(caar_state_norms(ctx, prm%np1, 0_c_int, nelemd, nrm), 'caar_state_norms')
  print *, "||v||_2  = ", nrm(1)
  print *, "||T||_2  = ", nrm(2)
  print *, "||dp||_2 = ", nrm(3)
  ! the downloaded host arrays hold the same state (spot value: element 1, level 1, point (1,1))
  print *, "T(1,1,1,np1,1)   = ", T(1,1,1,2,1)
  print *, "v(1,1,1,1,np1,1) = ", v(1,1,1,1,2,1)
  call caar_destroy(ctx)
end program caar_f90_driver
