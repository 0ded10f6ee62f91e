{-# LANGUAGE ForeignFunctionInterface #-}
-- | Device back end of "Data.Text.AhoCorasick.Searcher" (reference: src/Data/Text/AhoCorasick/Searcher.hs):
-- 'containsAny' (:156-164) and 'containsAll' (:173-187) for batches of haystacks.
--
-- A 'DeviceSearcher' is a reference 'Searcher.Searcher' plus its automaton in HBM; the reference's 'build',
-- 'buildWithValues', 'buildNeedleIdSearcher', 'needles', 'numNeedles', 'setCaseSensitivity' keep working on the
-- wrapped searcher.  'containsAll' needs 'machineValues' of the @Searcher Int@ in flat form (the needle ids a
-- state reports): the IntSet fold (:175-183) becomes one bitmap row per haystack on the device, and a haystack
-- whose set is empty is not scanned any further (@Done@, :181).
module Data.Text.AhoCorasick.Searcher.Device
  ( DeviceSearcher
  , toDevice
  , searcher
  , containsAny
  , containsAll
  ) where

import Data.Word (Word32, Word64, Word8)
import Foreign
import Foreign.C.Types

import qualified Data.Vector as Vector

import Data.Text.AhoCorasick.Automaton (AcMachine (..))
import Data.Text.Utf8 (Text)

import qualified Data.Text.AhoCorasick.Automaton.Device as Dev
import qualified Data.Text.AhoCorasick.Searcher as Searcher

data AmNeedleIds

foreign import ccall unsafe "am_needle_ids_create"
  c_am_needle_ids_create :: Ptr Dev.AmAutomaton -> Ptr Word64 -> Ptr Word32 -> Word32 -> Ptr (Ptr AmNeedleIds) -> IO CInt
foreign import ccall unsafe "&am_needle_ids_destroy"
  p_am_needle_ids_destroy :: FunPtr (Ptr AmNeedleIds -> IO ())
foreign import ccall safe "am_contains_all"
  c_am_contains_all :: Ptr AmNeedleIds -> CInt -> Ptr Dev.AmSlice -> CSize -> Ptr Word8 -> IO CInt

data DeviceSearcher v = DeviceSearcher
  { dsSearcher :: !(Searcher.Searcher v)
  , dsMachine  :: !(Dev.DeviceMachine v)
  }

-- | The reference searcher inside (for 'Searcher.needles', 'Searcher.numNeedles', 'Searcher.caseSensitivity', ...).
searcher :: DeviceSearcher v -> Searcher.Searcher v
searcher = dsSearcher

toDevice :: Searcher.Searcher v -> IO (DeviceSearcher v)
toDevice s = DeviceSearcher s <$> Dev.toDevice (Searcher.automaton s)

-- | 'Searcher.containsAny' (Searcher.hs:156-164) for a batch.
containsAny :: DeviceSearcher v -> [Text] -> IO [Bool]
containsAny (DeviceSearcher s dm) = Dev.containsAnyDevice (Searcher.caseSensitivity s) dm

-- | 'Searcher.containsAll' (Searcher.hs:173-187) for a batch, for searchers made by 'Searcher.buildNeedleIdSearcher' (:167-169).
containsAll :: DeviceSearcher Int -> [Text] -> IO [Bool]
containsAll (DeviceSearcher s dm) texts =
  let values  = machineValues (Dev.dmMachine dm)
      offsets = scanl (+) 0 (map (fromIntegral . length) (Vector.toList values)) :: [Word64]
      flat    = concatMap (map fromIntegral) (Vector.toList values) :: [Word32]
  in withArray offsets $ \pOff -> withArray (if null flat then [0] else flat) $ \pVals ->
     withForeignPtr (Dev.dmHandle dm) $ \ph -> alloca $ \out -> do
       c_am_needle_ids_create ph pOff pVals (fromIntegral (Searcher.numNeedles s)) out >>= Dev.checkRc
       ids <- peek out >>= newForeignPtr p_am_needle_ids_destroy
       Dev.withPinnedTexts texts $ \pSlices n ->
         allocaArray (max n 1) $ \pFlags -> withForeignPtr ids $ \pIds -> do
           c_am_contains_all pIds (Dev.caseFlag (Searcher.caseSensitivity s)) pSlices (fromIntegral n) pFlags >>= Dev.checkRc
           map (/= (0 :: Word8)) <$> peekArray n pFlags
