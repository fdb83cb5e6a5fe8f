! caar_mod.F90 -- Fortran (iso_c_binding) interface of libcaar_hip.so (include/caar.h).
!
! What the reference's Fortran driver would "use" to run its hot path
!     call compute_and_apply_rhs(np1,nm1,n0,qn0,dt2,elem,hvcoord,deriv,nets,nete,eta_ave_w)
!     (compute_and_apply_rhs_test/fortran/routine_mod.F90:7, called from main.F90:201-210)
! on the MI355X.  The element fields are handed over as flat Fortran-ordered arrays
!     v(np,np,2,nlev,timelevels,nelemd), T(np,np,nlev,timelevels,nelemd), ...
! i.e. elem(ie)%state%v etc. (element_state_mod.F90:17-23) gathered over ie; the
! library re-lays them out on the device (caar_upload_f90 / caar_download_f90).
module caar_mod
  use iso_c_binding
  implicit none
  public

  integer, parameter :: CAAR_OK = 0

  type, bind(C) :: caar_dims_t
    integer(c_int) :: np, nlev, qsize_d, timelevels, num_elems
  end type

  ! 16 pointers in the member order of CaarArrays / Homme::Arrays
  type, bind(C) :: caar_arrays_t
    type(c_ptr) :: elem_D, elem_Dinv, elem_fcor, elem_spheremp, elem_metdet, elem_rmetdet
    type(c_ptr) :: elem_state_dp3d, elem_state_v, elem_state_T, elem_state_phis, elem_state_Qdp
    type(c_ptr) :: elem_derived_eta_dot_dpdn, elem_derived_omega_p, elem_derived_phi
    type(c_ptr) :: elem_derived_pecnd, elem_derived_vn0
  end type

  ! 0-based time levels and element range [nets, nete); qn0 = -1 selects the dry branch
  type, bind(C) :: caar_params_t
    integer(c_int) :: nets, nete, n0, np1, nm1, qn0
    real(c_double) :: dt2, rrearth, eta_ave_w, Rwater_vapor, Rgas, kappa, ps0, hyai0
    type(c_ptr)    :: Dvv      ! np*np doubles, C order: Dvv_c((i-1)*np + j) = deriv%Dvv(i,j)
    integer(c_int) :: rsplit = 1          ! > 0: vertically Lagrangian (routine_mod.F90); 0: Eulerian
    type(c_ptr)    :: hybi = c_null_ptr     ! hvcoord%hybi(1:nlev+1), read when rsplit == 0
    type(c_ptr)    :: hybi_dev = c_null_ptr ! device copy for the stateless caar_launch only
  end type

  interface
    integer(c_int) function caar_supported(np, nlev) bind(C, name="caar_supported")
      import; integer(c_int), value :: np, nlev
    end function
    integer(c_int) function caar_device_count() bind(C, name="caar_device_count")
      import
    end function
    integer(c_int) function caar_create(ctx, dims, device) bind(C, name="caar_create")
      import; type(c_ptr) :: ctx; type(caar_dims_t) :: dims; integer(c_int), value :: device
    end function
    subroutine caar_destroy(ctx) bind(C, name="caar_destroy")
      import; type(c_ptr), value :: ctx
    end subroutine
    integer(c_int) function caar_upload_f90(ctx, f90_host, e0, e1) bind(C, name="caar_upload_f90")
      import; type(c_ptr), value :: ctx; type(caar_arrays_t) :: f90_host; integer(c_int), value :: e0, e1
    end function
    integer(c_int) function caar_upload_f90_arrays(ctx, f90_host, e0, e1, array_mask) bind(C, name="caar_upload_f90_arrays")
      import; type(c_ptr), value :: ctx; type(caar_arrays_t) :: f90_host; integer(c_int), value :: e0, e1, array_mask
    end function
    integer(c_int) function caar_download_f90(ctx, f90_host, e0, e1, all_arrays) bind(C, name="caar_download_f90")
      import; type(c_ptr), value :: ctx; type(caar_arrays_t) :: f90_host; integer(c_int), value :: e0, e1, all_arrays
    end function
    integer(c_int) function caar_run(ctx, prm) bind(C, name="caar_run")
      import; type(c_ptr), value :: ctx; type(caar_params_t) :: prm
    end function
    ! nsteps calls with update_time_levels between them (main.F90:201-210 with its rotation) as one launch: the caller
    ! rotates its own np1 / nm1 / n0 nsteps times afterwards
    integer(c_int) function caar_run_steps(ctx, prm, nsteps, rotate) bind(C, name="caar_run_steps")
      import; type(c_ptr), value :: ctx; type(caar_params_t) :: prm; integer(c_int), value :: nsteps, rotate
    end function
    integer(c_int) function caar_sync(ctx) bind(C, name="caar_sync")
      import; type(c_ptr), value :: ctx
    end function
    integer(c_int) function caar_state_norms(ctx, tl, e0, e1, out) bind(C, name="caar_state_norms")
      import; type(c_ptr), value :: ctx; integer(c_int), value :: tl, e0, e1; real(c_double) :: out(3)
    end function
  end interface

contains

  subroutine caar_check(rc, what)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: what
    if (rc /= CAAR_OK) then
      print *, 'caar: ', what, ' failed with code ', rc
      error stop 1
    end if
  end subroutine

end module caar_mod
